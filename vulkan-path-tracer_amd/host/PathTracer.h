// PathTracer.h — the reference's render API (PathTracer/PathTracer.h:83-183) over the MI355X backend's C-ABI
// (include/vpt.h).  Same member names and meanings; what changes is what the Vulkan types carried:
//   * New(device ordinal) instead of New(VulkanHelper::Device, ThreadPool*);
//   * no CommandBuffer arguments — every call completes before it returns;
//   * GetOutputImage() hands back the RGBA32F accumulation image as host floats;
//   * the seed is explicit (SetSeed): the reference draws it from the wall clock (PathTracer.cpp:127-140);
//   * errors throw std::runtime_error where the reference VH_ASSERT-aborts.
// Box volumes (AddVolume / SetVolume / RemoveVolume / SetPhaseFunction) and the atmosphere members (SetEnableAtmosphere, SetPlanetRadius,
// ...) are in; heterogeneous volumes take their density as a dense grid of decoded voxels (the C-ABI's vpt_add_density_grid): reading
// .vdb / NanoVDB files (AddDensityDataToVolume, PathTracer.cpp:1347-1516) is the one member that is absent.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vpt.h"
#include "Math.h"
#include "SceneLoader.h"

namespace vpthost {

class PathTracer {
public:
    using Material = vpt_material;  // byte-identical to PathTracer::Material (PathTracer.h:12-34)
    struct Volume {  // PathTracer.h:36-74 without the NanoVDB buffers
        Vec3 CornerMin{-1.0f, -1.0f, -1.0f}, CornerMax{1.0f, 1.0f, 1.0f}, Position{0.0f, 0.0f, 0.0f}, Scale{1.0f, 1.0f, 1.0f};
        Vec3 Color{0.8f, 0.8f, 0.8f}, EmissiveColor{0.0f, 0.0f, 0.0f};
        float Density = 1.0f, Anisotropy = 0.0f, Alpha = 1.0f, DropletSize = 20.0f;
        int ApproximatedScatteringForClouds = 0;
        float ApproximatedScatteringFalloff = 0.8f;
        int DensityDataIndex = -1;   // set by AddDensityDataToVolume
        float GridSharpness = 1.0f;
        Vec3 TemperatureColor{1.0f, 0.5f, 0.0f};
        int UseBlackbody = 1, HasTemperatureData = 0, KelvinMin = 500, KelvinMax = 8000;
        float TemperatureGamma = 1.0f, TemperatureScale = 1.0f, EmissiveColorGamma = 1.0f;
    };
    enum class PhaseFunction { HENYEY_GREENSTEIN = 0, DRAINE = 1, HENYEY_GREENSTEIN_PLUS_DRAINE = 2 };  // PathTracer.h:76-81

    [[nodiscard]] static PathTracer New(int device = 0);
    // Multi-GPU: shard `shardRank` of `shardCount` renders rows y % shardCount == shardRank on `device` (the
    // reference's interleaved split-screen partition, RayGen.slang:16-25, along one axis).  GatherShards() is the one
    // collective of the path: every shard's rows go to shards[root] over xGMI (vpt_multi_gather_shards), after which
    // shards[root].GetOutputImage() / PostProcessor see the whole image — bit-identical to a single-device render.
    // framesInFlight / residentFrames: vpt_config.frames_in_flight / resident_frames — the device-memory cap of an offline host (0, 0 = the library's
    // default schedule: 126 GB on a 1080p scene whose BVH lives in memory; 64, 16 = 14.4 GB; INTEGRATION.md "Device memory").  Images do not depend on them.
    [[nodiscard]] static PathTracer New(int device, uint32_t shardRank, uint32_t shardCount, uint32_t framesInFlight = 0, uint32_t residentFrames = 0);
    static void GatherShards(const std::vector<PathTracer*>& shards, uint32_t root = 0);
    [[nodiscard]] uint32_t GetShardRank() const { return m_ShardRank; }
    [[nodiscard]] uint32_t GetShardCount() const { return m_ShardCount; }
    PathTracer() = default;
    PathTracer(PathTracer&& o) noexcept;
    PathTracer& operator=(PathTracer&& o) noexcept;
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;
    ~PathTracer();

    // PathTracer.cpp:158-676. `lookupTablePath`: the three energy-compensation tables as one raw file.
    void SetScene(const std::string& sceneFilePath);
    void SetScene(const SceneAsset& scene);
    void SetLookupTablePath(const std::string& path) { m_LookupTablePath = path; }
    // True when all samples were accumulated (PathTracer.cpp:122-156). `dispatches` > 1 lets the backend keep
    // several frames in flight; PathTrace() == one reference call.
    bool PathTrace(uint32_t dispatches = 1);
    // The same as the reference has it — the dispatch is RECORDED and the call returns (PathTracer.cpp:122-156 records into the frame's
    // command buffer): vpt_render_async.  The returned ticket is the fence value of that work; Wait(ticket) blocks until it has finished
    // (0: everything enqueued so far).  Every member that reads or changes device state drains outstanding work first.
    bool PathTraceAsync(uint32_t dispatches = 1, uint64_t* ticket = nullptr);
    void Wait(uint64_t ticket = 0);
    void ResizeImage(uint32_t width, uint32_t height);

    [[nodiscard]] const std::vector<float>& GetOutputImage();  // RGBA32F, width*height*4
    [[nodiscard]] uint32_t GetWidth() const { return m_Width; }
    [[nodiscard]] uint32_t GetHeight() const { return m_Height; }

    [[nodiscard]] const std::vector<Material>& GetMaterials() const { return m_Materials; }
    [[nodiscard]] const Material& GetMaterial(uint32_t index) const { return m_Materials[index]; }
    [[nodiscard]] const std::string& GetMaterialName(uint32_t index) const { return m_MaterialNames[index]; }
    void SetMaterial(uint32_t index, const Material& material);

    void SetSkyMIS(bool value) { SetFlag(VPT_FLAG_SKY_MIS, value); }
    void SetMeshMIS(bool value) { SetFlag(VPT_FLAG_MESH_MIS, value); }
    void SetEnvMapShownDirectly(bool value) { SetFlag(VPT_FLAG_SHOW_ENV_DIRECTLY, value); }
    void SetUseOnlyGeometryNormals(bool value) { SetFlag(VPT_FLAG_GEOMETRY_NORMALS, value); }
    void SetUseEnergyCompensation(bool value) { SetFlag(VPT_FLAG_ENERGY_COMPENSATION, value); }
    void SetFurnaceTestMode(bool value) { SetFlag(VPT_FLAG_FURNACE, value); }
    void SetUseRayQueries(bool value);
    // PathTracer.h:106,157-159.  Volumes survive SetScene, like m_Volumes upstream.
    void AddVolume(const Volume& volume);
    void RemoveVolume(uint32_t index);
    void SetVolume(uint32_t index, const Volume& volume);
    void SetPhaseFunction(PhaseFunction phaseFunction);
    // AddDensityDataToVolume (PathTracer.h:165, PathTracer.cpp:1347-1516) with the .vdb already decoded into a dense grid of
    // raw densities (x fastest, file index order).  The box corners stay as the caller set them.
    void AddDensityDataToVolume(uint32_t volumeIndex, uint32_t dimX, uint32_t dimY, uint32_t dimZ, const float* density);
    void RemoveDensityDataFromVolume(uint32_t volumeIndex);
    [[nodiscard]] uint32_t GetVolumesCount() const { return (uint32_t)m_Volumes.size(); }
    [[nodiscard]] const std::vector<Volume>& GetVolumes() const { return m_Volumes; }
    [[nodiscard]] PhaseFunction GetPhaseFunction() const { return m_PhaseFunction; }
    // Atmosphere (PathTracer.h:133-144, 168-179; defaults :221-232).  The sun direction is SkyAzimuth / SkyAltitude.
    void SetEnableAtmosphere(bool enable) { m_EnableAtmosphere = enable; UploadAtmosphere(); }
    void SetPlanetPosition(Vec3 p) { m_Atmosphere.planet_position[0] = p.x; m_Atmosphere.planet_position[1] = p.y; m_Atmosphere.planet_position[2] = p.z; UploadAtmosphere(); }
    void SetPlanetRadius(float v) { m_Atmosphere.planet_radius = v; UploadAtmosphere(); }
    void SetAtmosphereHeight(float v) { m_Atmosphere.atmosphere_height = v; UploadAtmosphere(); }
    void SetRayleighScatteringCoefficientMultiplier(Vec3 m) { Set3(m_Atmosphere.rayleigh_multiplier, m); }
    void SetMieScatteringCoefficientMultiplier(Vec3 m) { Set3(m_Atmosphere.mie_multiplier, m); }
    void SetOzoneAbsorptionCoefficientMultiplier(Vec3 m) { Set3(m_Atmosphere.ozone_multiplier, m); }
    void SetRayleighDensityFalloff(float v) { m_Atmosphere.rayleigh_density_falloff = v; UploadAtmosphere(); }
    void SetMieDensityFalloff(float v) { m_Atmosphere.mie_density_falloff = v; UploadAtmosphere(); }
    void SetOzoneDensityFalloff(float v) { m_Atmosphere.ozone_density_falloff = v; UploadAtmosphere(); }
    void SetOzonePeak(float v) { m_Atmosphere.ozone_peak = v; UploadAtmosphere(); }
    void SetSunColor(Vec3 c) { Set3(m_Atmosphere.sun_color, c); }
    [[nodiscard]] bool IsAtmosphereEnabled() const { return m_EnableAtmosphere; }
    [[nodiscard]] const vpt_atmosphere& GetAtmosphere() const { return m_Atmosphere; }
    [[nodiscard]] Vec3 GetPlanetPosition() const { return Vec3(m_Atmosphere.planet_position[0], m_Atmosphere.planet_position[1], m_Atmosphere.planet_position[2]); }
    [[nodiscard]] float GetPlanetRadius() const { return m_Atmosphere.planet_radius; }
    [[nodiscard]] float GetAtmosphereHeight() const { return m_Atmosphere.atmosphere_height; }
    [[nodiscard]] Vec3 GetRayleighScatteringCoefficientMultiplier() const { return Vec3(m_Atmosphere.rayleigh_multiplier[0], m_Atmosphere.rayleigh_multiplier[1], m_Atmosphere.rayleigh_multiplier[2]); }
    [[nodiscard]] Vec3 GetMieScatteringCoefficientMultiplier() const { return Vec3(m_Atmosphere.mie_multiplier[0], m_Atmosphere.mie_multiplier[1], m_Atmosphere.mie_multiplier[2]); }
    [[nodiscard]] Vec3 GetOzoneAbsorptionCoefficientMultiplier() const { return Vec3(m_Atmosphere.ozone_multiplier[0], m_Atmosphere.ozone_multiplier[1], m_Atmosphere.ozone_multiplier[2]); }
    [[nodiscard]] float GetRayleighDensityFalloff() const { return m_Atmosphere.rayleigh_density_falloff; }
    [[nodiscard]] float GetMieDensityFalloff() const { return m_Atmosphere.mie_density_falloff; }
    [[nodiscard]] float GetOzoneDensityFalloff() const { return m_Atmosphere.ozone_density_falloff; }
    [[nodiscard]] float GetOzonePeak() const { return m_Atmosphere.ozone_peak; }
    [[nodiscard]] Vec3 GetSunColor() const { return Vec3(m_Atmosphere.sun_color[0], m_Atmosphere.sun_color[1], m_Atmosphere.sun_color[2]); }
    // GetOutputImageView() (PathTracer.h:112) is a Vulkan image view upstream; here GetOutputImage() returns the RGBA32F host copy.
    // ReloadShaders (PathTracer.h:92, called from Editor.cpp:437 behind the "Reload Shaders" button): upstream recompiles the Slang sources and rebuilds
    // the ray-tracing pipeline.  Here every kernel is compiled into libvpt_hip.so and feature toggles are flag bits, so there is nothing to
    // reload: a callable no-op that keeps the accumulated image (upstream's reload does not reset it either), so an unmodified Editor.cpp links.
    void ReloadShaders() {}
    void SetCameraViewInverse(const Mat4& view);
    void SetCameraProjectionInverse(const Mat4& projection);
    void SetMaxSamplesAccumulated(uint32_t v) { m_Params.max_samples = v; Push(false); }
    void SetMaxDepth(uint32_t v) { m_Params.max_depth = v; Push(true); }
    void SetSamplesPerFrame(uint32_t v) { m_Params.samples_per_frame = v; Push(true); }
    void SetMaxLuminance(float v) { m_Params.max_luminance = v; Push(true); }
    void SetFocusDistance(float v) { m_Params.focus_distance = v; Push(true); }
    void SetDepthOfFieldStrength(float v) { m_Params.dof_strength = v; Push(true); }
    void SetSkyAzimuth(float v) { m_Params.sky_azimuth = v; Push(true); }
    void SetSkyAltitude(float v) { m_Params.sky_altitude = v; Push(true); }
    void SetSkyIntensity(float v) { m_Params.sky_intensity = v; Push(true); }
    void SetSplitScreenCount(uint32_t v) { m_Params.screen_chunk_count = v; Push(true); }
    void SetEmissiveMeshSamplingPDFBias(float v) { m_Params.emissive_pdf_bias = v; Push(true); }
    void SetSeed(uint32_t v) { m_Params.base_seed = v; Push(true); }
    // RGBA32F, width*height*4 floats (the reference loads a .hdr file: SetEnvMapFilepath, PathTracer.cpp:1137-1164)
    void SetEnvironmentMap(const std::vector<float>& rgba, uint32_t width, uint32_t height);
    void SetEnvMapFilepath(const std::string& filePath);  // PathTracer.h:154: a Radiance .hdr file
    [[nodiscard]] const std::string& GetEnvMapFilepath() const { return m_EnvMapFilepath; }

    [[nodiscard]] uint32_t GetSamplesAccumulated() const { return m_SamplesAccumulated; }
    [[nodiscard]] uint32_t GetSamplesPerFrame() const { return m_Params.samples_per_frame; }
    [[nodiscard]] uint32_t GetMaxSamplesAccumulated() const { return m_Params.max_samples; }
    [[nodiscard]] uint32_t GetMaxDepth() const { return m_Params.max_depth; }
    [[nodiscard]] float GetMaxLuminance() const { return m_Params.max_luminance; }
    [[nodiscard]] float GetFocusDistance() const { return m_Params.focus_distance; }
    [[nodiscard]] float GetDepthOfFieldStrength() const { return m_Params.dof_strength; }
    [[nodiscard]] float GetSkyRotationAzimuth() const { return m_Params.sky_azimuth; }
    [[nodiscard]] float GetSkyRotationAltitude() const { return m_Params.sky_altitude; }
    [[nodiscard]] float GetSkyIntensity() const { return m_Params.sky_intensity; }
    [[nodiscard]] bool IsSkyMISEnabled() const { return m_Params.flags & VPT_FLAG_SKY_MIS; }
    [[nodiscard]] bool IsMeshMISEnabled() const { return m_Params.flags & VPT_FLAG_MESH_MIS; }
    [[nodiscard]] bool IsEnvMapShownDirectly() const { return m_Params.flags & VPT_FLAG_SHOW_ENV_DIRECTLY; }
    [[nodiscard]] bool UseOnlyGeometryNormals() const { return m_Params.flags & VPT_FLAG_GEOMETRY_NORMALS; }
    [[nodiscard]] bool UseEnergyCompensation() const { return m_Params.flags & VPT_FLAG_ENERGY_COMPENSATION; }
    [[nodiscard]] bool IsInFurnaceTestMode() const { return m_Params.flags & VPT_FLAG_FURNACE; }
    [[nodiscard]] bool UseRayQueries() const { return (m_Params.flags & VPT_FLAG_RAY_QUERIES) != 0; }
    [[nodiscard]] uint32_t GetSplitScreenCount() const { return m_Params.screen_chunk_count; }
    [[nodiscard]] float GetEmissiveMeshSamplingPDFBias() const { return m_Params.emissive_pdf_bias; }
    [[nodiscard]] const Mat4& GetCameraViewInverse() const { return m_CameraViewInverse; }
    [[nodiscard]] const Mat4& GetCameraProjectionInverse() const { return m_CameraProjectionInverse; }
    [[nodiscard]] uint64_t GetTotalVertexCount() const { return m_TotalVertexCount; }
    [[nodiscard]] uint64_t GetTotalIndexCount() const { return m_TotalIndexCount; }

    void ResetPathTracing();  // PathTracer.h:183

    [[nodiscard]] vpt_ctx* Context() const { return m_Ctx; }  // for PostProcessor

private:
    void Swap(PathTracer& o) noexcept;
    void Check(int rc, const char* what) const;
    void SetFlag(uint32_t bit, bool value);
    void Push(bool resets);
    void UploadScene();
    void UploadVolumes();
    void UploadAtmosphere();
    void Set3(float* dst, Vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; UploadAtmosphere(); }

    int m_Device = 0;
    uint32_t m_ShardRank = 0, m_ShardCount = 1;
    uint32_t m_FramesInFlight = 0, m_ResidentFrames = 0;
    vpt_ctx* m_Ctx = nullptr;
    vpt_params m_Params{};
    uint32_t m_Width = 0, m_Height = 0;
    uint32_t m_SamplesAccumulated = 0;
    uint64_t m_DispatchCount = 0;
    uint64_t m_TotalVertexCount = 0, m_TotalIndexCount = 0;
    Mat4 m_CameraViewInverse, m_CameraProjectionInverse;
    std::vector<Material> m_Materials;
    std::vector<std::string> m_MaterialNames;
    SceneAsset m_Scene;
    std::vector<float> m_Env; uint32_t m_EnvW = 1, m_EnvH = 1;
    std::vector<float> m_LutR, m_LutO, m_LutI;
    std::string m_LookupTablePath;
    std::string m_EnvMapFilepath;
    std::vector<float> m_Output;
    std::vector<Volume> m_Volumes;
    bool m_EnableAtmosphere = false;  // PathTracer.h:221
    vpt_atmosphere m_Atmosphere = [] { vpt_atmosphere a; vpt_default_atmosphere(&a); return a; }();
    PhaseFunction m_PhaseFunction = PhaseFunction::HENYEY_GREENSTEIN;  // PathTracer.h:219
};

}  // namespace vpthost
