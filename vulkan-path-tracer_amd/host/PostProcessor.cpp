#include "PostProcessor.h"

#include <stdexcept>
#include <string>

namespace vpthost {

void PostProcessor::PostProcess() {
    if (!m_Source || !m_Source->Context()) throw std::runtime_error("PostProcess: SetInputImage with a PathTracer that has a scene first");
    vpt_post_params pp{m_Tonemap.Exposure, m_Tonemap.Gamma, m_Bloom.BloomThreshold, m_Bloom.BloomStrength, m_Bloom.MipCount, m_Bloom.FalloffRange};
    m_Output.resize((size_t)m_Source->GetWidth() * m_Source->GetHeight() * 4);
    int rc = vpt_postprocess(m_Source->Context(), &pp, m_Output.data(), nullptr);
    if (rc != VPT_OK) throw std::runtime_error(std::string("vpt_postprocess failed: ") + vpt_last_error(m_Source->Context()));
}

uint64_t PostProcessor::PostProcessAsync(void* rgba8Device) {
    if (!m_Source || !m_Source->Context()) throw std::runtime_error("PostProcess: SetInputImage with a PathTracer that has a scene first");
    vpt_post_params pp{m_Tonemap.Exposure, m_Tonemap.Gamma, m_Bloom.BloomThreshold, m_Bloom.BloomStrength, m_Bloom.MipCount, m_Bloom.FalloffRange};
    uint64_t ticket = 0;
    int rc = vpt_postprocess_device(m_Source->Context(), &pp, rgba8Device, &ticket);
    if (rc != VPT_OK) throw std::runtime_error(std::string("vpt_postprocess_device failed: ") + vpt_last_error(m_Source->Context()));
    return ticket;
}
const void* PostProcessor::GetOutputImageView() const { return m_Source && m_Source->Context() ? vpt_output_device(m_Source->Context()) : nullptr; }

}  // namespace vpthost
