// PostProcessor.h — PathTracer/PostProcessor.h:8-33 over the C-ABI: bloom chain + ACES tonemap of a PathTracer's
// accumulation image (PostProcessor.cpp:193-246).  GetOutputImage() is the RGBA8 image the Editor displays / saves.
#pragma once
#include <cstdint>
#include <vector>

#include "PathTracer.h"

namespace vpthost {

class PostProcessor {
public:
    struct TonemappingData { float Exposure = 1.0f; float Gamma = 2.2f; };
    struct BloomData { float BloomThreshold = 2.0f; float BloomStrength = 1.0f; uint32_t MipCount = 10; float FalloffRange = 5.0f; };

    static PostProcessor New() { return PostProcessor(); }
    void SetInputImage(PathTracer& source) { m_Source = &source; }
    void PostProcess();
    // PostProcess(cmd) as the reference records it (PostProcessor.cpp:193-246): enqueued behind the source's renders, nothing waited for;
    // the RGBA8 image stays on the device — GetOutputImageView() is its pointer (PostProcessor.h GetOutputImageView upstream), or pass
    // an interop / swapchain image as `rgba8Device`.  Returns the ticket to hand to PathTracer::Wait.
    uint64_t PostProcessAsync(void* rgba8Device = nullptr);
    [[nodiscard]] const void* GetOutputImageView() const;
    // ReloadShaders (PostProcessor.h:31, Editor.cpp:438): the bloom / tonemap kernels live in libvpt_hip.so — a callable no-op (see PathTracer::ReloadShaders)
    void ReloadShaders() {}
    void SetTonemappingData(const TonemappingData& data) { m_Tonemap = data; }
    void SetBloomData(const BloomData& data) { m_Bloom = data; }
    [[nodiscard]] const std::vector<uint8_t>& GetOutputImage() const { return m_Output; }  // RGBA8 UNORM

private:
    PathTracer* m_Source = nullptr;
    TonemappingData m_Tonemap;
    BloomData m_Bloom;
    std::vector<uint8_t> m_Output;
};

}  // namespace vpthost
