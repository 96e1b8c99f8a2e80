// LookupTableCalculator.h — PathTracer/LookupTableCalculator.h:5-34 over the C-ABI (vpt_lut_calculate).
// The reference selects the table by the compute shader it compiles ("LookupReflect.slang", or
// "LookupRefract.slang" with the define ABOVE_SURFACE / BELOW_SURFACE, Application.cpp:40,53,66); the same
// strings select the HIP kernel here.  CalculateTable(size, sampleCount) == LookupTableCalculator.cpp:44-157.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vpthost {

struct ShaderDefine { std::string Name, Value; };  // VulkanHelper::Shader::Define
struct UVec3 { uint32_t x, y, z; };                // glm::uvec3

class LookupTableCalculator {
public:
    // `device` = HIP device index (the reference passes its VulkanHelper::Device).  Throws std::runtime_error on an
    // unknown shader / define combination.
    [[nodiscard]] static LookupTableCalculator New(int device, const std::string& shaderFilepath, const std::vector<ShaderDefine>& defines);
    // Throws std::runtime_error if sampleCount < 20 (not one full pass) or on a device error.
    std::vector<float> CalculateTable(UVec3 tableSize, uint32_t sampleCount);
    // The reference seeds each pass from the wall clock (LookupTableCalculator.cpp:101-102); here the clock reading
    // is a settable constant so tables are reproducible.  Default 0.
    void SetTimeSeed(uint32_t timeMillis) { m_TimeMillis = timeMillis; }
    void Destroy() {}

private:
    int m_Device = 0;
    uint32_t m_Kind = 0;
    uint32_t m_TimeMillis = 0;
};

}  // namespace vpthost
