#include "LookupTableCalculator.h"

#include <stdexcept>

#include "../../include/vpt.h"

namespace vpthost {

LookupTableCalculator LookupTableCalculator::New(int device, const std::string& shaderFilepath, const std::vector<ShaderDefine>& defines) {
    LookupTableCalculator c;
    c.m_Device = device;
    auto ends_with = [&](const char* tail) {
        const std::string t(tail);
        return shaderFilepath.size() >= t.size() && shaderFilepath.compare(shaderFilepath.size() - t.size(), t.size(), t) == 0;
    };
    bool above = false, below = false;
    for (const ShaderDefine& d : defines) { above |= d.Name == "ABOVE_SURFACE"; below |= d.Name == "BELOW_SURFACE"; }
    if (ends_with("LookupReflect.slang")) c.m_Kind = VPT_LUT_REFLECT;
    else if (ends_with("LookupRefract.slang")) c.m_Kind = above ? VPT_LUT_REFRACT_ABOVE : VPT_LUT_REFRACT_BELOW;  // #ifdef ABOVE_SURFACE ... #else
    else throw std::runtime_error("LookupTableCalculator: unknown shader " + shaderFilepath);
    if (above && below) throw std::runtime_error("LookupTableCalculator: ABOVE_SURFACE and BELOW_SURFACE are exclusive");
    return c;
}

std::vector<float> LookupTableCalculator::CalculateTable(UVec3 size, uint32_t sampleCount) {
    std::vector<float> result((size_t)size.x * size.y * size.z, 0.0f);
    int rc = vpt_lut_calculate(m_Device, m_Kind, size.x, size.y, size.z, sampleCount, m_TimeMillis, result.data());
    if (rc != VPT_OK) throw std::runtime_error("vpt_lut_calculate failed with code " + std::to_string(rc));
    return result;
}

}  // namespace vpthost
