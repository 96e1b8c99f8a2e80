// PathTracer.cpp — forwards the reference's PathTracer members to the C-ABI (see PathTracer.h).
#include "PathTracer.h"

#include <algorithm>

#include <cstring>
#include <stdexcept>
#include <utility>

namespace vpthost {

void PathTracer::Check(int rc, const char* what) const {
    if (rc != VPT_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (m_Ctx ? vpt_last_error(m_Ctx) : ""));
}

PathTracer PathTracer::New(int device) {
    PathTracer pt;
    pt.m_Device = device;
    vpt_default_params(&pt.m_Params);
    pt.m_Env.assign(4, 0.0f);  // black 1x1 environment until SetEnvironmentMap (the reference's default .hdr is not redistributable)
    return pt;
}
PathTracer PathTracer::New(int device, uint32_t shardRank, uint32_t shardCount, uint32_t framesInFlight, uint32_t residentFrames) {
    if (shardCount == 0 || shardRank >= shardCount) throw std::runtime_error("PathTracer::New: shardRank must be < shardCount");
    PathTracer pt = New(device);
    pt.m_ShardRank = shardRank; pt.m_ShardCount = shardCount;
    pt.m_FramesInFlight = framesInFlight; pt.m_ResidentFrames = residentFrames;
    return pt;
}
void PathTracer::GatherShards(const std::vector<PathTracer*>& shards, uint32_t root) {
    if (shards.empty() || root >= shards.size()) throw std::runtime_error("GatherShards: bad arguments");
    std::vector<vpt_ctx*> ctxs;
    for (PathTracer* p : shards) { if (!p || !p->m_Ctx) throw std::runtime_error("GatherShards before SetScene"); ctxs.push_back(p->m_Ctx); }
    shards[root]->Check(vpt_multi_gather_shards(ctxs.data(), (uint32_t)ctxs.size(), root), "vpt_multi_gather_shards");
}
// Member-wise swap of EVERY member (a hand-written member list once forgot the volume / atmosphere state): the moved-from
// object ends up with the target's old state and releases it in its own destructor.
void PathTracer::Swap(PathTracer& o) noexcept {
    using std::swap;
    swap(m_Device, o.m_Device); swap(m_ShardRank, o.m_ShardRank); swap(m_ShardCount, o.m_ShardCount); swap(m_Ctx, o.m_Ctx);
    swap(m_FramesInFlight, o.m_FramesInFlight); swap(m_ResidentFrames, o.m_ResidentFrames);
    swap(m_Params, o.m_Params); swap(m_Width, o.m_Width); swap(m_Height, o.m_Height);
    swap(m_SamplesAccumulated, o.m_SamplesAccumulated); swap(m_DispatchCount, o.m_DispatchCount);
    swap(m_TotalVertexCount, o.m_TotalVertexCount); swap(m_TotalIndexCount, o.m_TotalIndexCount);
    swap(m_CameraViewInverse, o.m_CameraViewInverse); swap(m_CameraProjectionInverse, o.m_CameraProjectionInverse);
    swap(m_Materials, o.m_Materials); swap(m_MaterialNames, o.m_MaterialNames); swap(m_Scene, o.m_Scene);
    swap(m_Env, o.m_Env); swap(m_EnvW, o.m_EnvW); swap(m_EnvH, o.m_EnvH);
    swap(m_LutR, o.m_LutR); swap(m_LutO, o.m_LutO); swap(m_LutI, o.m_LutI);
    swap(m_LookupTablePath, o.m_LookupTablePath); swap(m_EnvMapFilepath, o.m_EnvMapFilepath); swap(m_Output, o.m_Output);
    swap(m_Volumes, o.m_Volumes); swap(m_EnableAtmosphere, o.m_EnableAtmosphere); swap(m_Atmosphere, o.m_Atmosphere);
    swap(m_PhaseFunction, o.m_PhaseFunction);
}
PathTracer::PathTracer(PathTracer&& o) noexcept { Swap(o); }
PathTracer& PathTracer::operator=(PathTracer&& o) noexcept {
    if (this != &o) {
        PathTracer released;   // takes this object's old state (context included) and frees it on scope exit
        Swap(released);
        Swap(o);
    }
    return *this;
}
PathTracer::~PathTracer() { if (m_Ctx) vpt_destroy(m_Ctx); }

void PathTracer::SetScene(const std::string& sceneFilePath) {
    SceneAsset scene; std::string err;
    if (!ImportScene(sceneFilePath, scene, err)) throw std::runtime_error("Failed to import scene! " + err);  // PathTracer.cpp:168
    SetScene(scene);
}

void PathTracer::SetScene(const SceneAsset& sceneIn) {
    ResetPathTracing();
    m_Scene = sceneIn;
    if (m_Scene.Cameras.empty()) {  // PathTracer.cpp:171-178
        CameraAsset cam;
        cam.AspectRatio = 16.0f / 9.0f; cam.FOV = 45.0f;
        cam.ViewMatrix = lookAt(Vec3(0.0f, 0.0f, 5.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f));
        m_Scene.Cameras.push_back(cam);
    }
    if (m_Scene.Meshes.empty()) throw std::runtime_error("No meshes found in scene! Please load a scene that contains meshes!");
    if (m_Scene.Meshes.size() >= VPT_MAX_ENTITIES || m_Scene.Materials.size() >= VPT_MAX_ENTITIES || m_Scene.MeshInstances.size() >= VPT_MAX_INSTANCES)
        throw std::runtime_error("Too many meshes / materials / mesh instances in the scene");  // PathTracer.cpp:182-184
    const float aspectRatio = m_Scene.Cameras[0].AspectRatio;
    m_CameraViewInverse = inverse(m_Scene.Cameras[0].ViewMatrix);                             // PathTracer.cpp:188
    m_CameraProjectionInverse = inverse(perspective(radians(45.0f), aspectRatio, 0.1f, 100.0f));  // PathTracer.cpp:578: the camera's own FOV is ignored
    if (m_LutR.empty()) {
        std::string err;
        if (m_LookupTablePath.empty()) throw std::runtime_error("SetLookupTablePath() must name the energy-compensation tables before SetScene()");
        if (!LoadLookupTables(m_LookupTablePath, m_LutR, m_LutO, m_LutI, err)) throw std::runtime_error(err);
    }
    m_Materials = m_Scene.Materials; m_MaterialNames = m_Scene.MaterialNames;
    m_TotalVertexCount = 0; m_TotalIndexCount = 0;
    for (const auto& m : m_Scene.Meshes) { m_TotalVertexCount += m.Vertices.size(); m_TotalIndexCount += m.Indices.size(); }
    const uint32_t w = (uint32_t)(1080.0f * aspectRatio), h = 1080;  // PathTracer.cpp:509-511
    if (m_Width == 0) { m_Width = w; m_Height = h; }  // unless ResizeImage already chose a size
    if (!m_Ctx) {
        vpt_config cfg{}; cfg.device = m_Device; cfg.width = m_Width; cfg.height = m_Height; cfg.shard_rank = m_ShardRank; cfg.shard_count = m_ShardCount;
        cfg.frames_in_flight = m_FramesInFlight; cfg.resident_frames = m_ResidentFrames;
        int err = 0;
        m_Ctx = vpt_create(&cfg, &err);
        if (!m_Ctx) throw std::runtime_error("vpt_create failed (" + std::to_string(err) + "): no usable HIP device; this backend has no CPU fallback");
    }
    UploadScene();
    Check(vpt_set_params(m_Ctx, &m_Params), "vpt_set_params");
    Check(vpt_set_camera(m_Ctx, m_CameraViewInverse.m, m_CameraProjectionInverse.m), "vpt_set_camera");
    UploadVolumes();
    UploadAtmosphere();
}

// VolumeGPU(const Volume&), PathTracer.h:374-399: the box goes to world space on upload.
void PathTracer::UploadVolumes() {
    if (!m_Ctx) return;
    std::vector<vpt_volume> g;
    for (const Volume& v : m_Volumes) {
        vpt_volume o{};
        const float mn[3] = {v.Position.x + v.CornerMin.x * v.Scale.x, v.Position.y + v.CornerMin.y * v.Scale.y, v.Position.z + v.CornerMin.z * v.Scale.z};
        const float mx[3] = {v.Position.x + v.CornerMax.x * v.Scale.x, v.Position.y + v.CornerMax.y * v.Scale.y, v.Position.z + v.CornerMax.z * v.Scale.z};
        std::memcpy(o.corner_min, mn, 12); std::memcpy(o.corner_max, mx, 12);
        o.color[0] = v.Color.x; o.color[1] = v.Color.y; o.color[2] = v.Color.z;
        o.emissive_color[0] = v.EmissiveColor.x; o.emissive_color[1] = v.EmissiveColor.y; o.emissive_color[2] = v.EmissiveColor.z;
        o.density = v.Density; o.anisotropy = v.Anisotropy; o.alpha = v.Alpha; o.droplet_size = v.DropletSize;
        o.density_data_index = v.DensityDataIndex; o.grid_sharpness = v.GridSharpness;
        o.has_temperature_data = v.HasTemperatureData; o.use_blackbody = v.UseBlackbody;
        o.temperature_color[0] = v.TemperatureColor.x; o.temperature_color[1] = v.TemperatureColor.y; o.temperature_color[2] = v.TemperatureColor.z;
        o.temperature_gamma = v.TemperatureGamma; o.temperature_scale = v.TemperatureScale; o.emissive_color_gamma = v.EmissiveColorGamma;
        o.kelvin_min = v.KelvinMin; o.kelvin_max = v.KelvinMax;
        o.approximated_scattering = v.ApproximatedScatteringForClouds; o.approximated_scattering_falloff = v.ApproximatedScatteringFalloff;
        g.push_back(o);
    }
    Check(vpt_set_phase_function(m_Ctx, (uint32_t)m_PhaseFunction), "vpt_set_phase_function");
    Check(vpt_set_volumes(m_Ctx, g.data(), (uint32_t)g.size()), "vpt_set_volumes");
    m_SamplesAccumulated = 0; m_DispatchCount = 0;
}
void PathTracer::UploadAtmosphere() {
    if (!m_Ctx) return;
    Check(vpt_set_atmosphere(m_Ctx, m_EnableAtmosphere ? &m_Atmosphere : nullptr), "vpt_set_atmosphere");
    m_SamplesAccumulated = 0; m_DispatchCount = 0;
}
void PathTracer::AddVolume(const Volume& volume) { m_Volumes.push_back(volume); UploadVolumes(); }
void PathTracer::RemoveVolume(uint32_t index) {
    if (index >= m_Volumes.size()) throw std::runtime_error("RemoveVolume: index out of range");
    m_Volumes.erase(m_Volumes.begin() + index); UploadVolumes();
}
void PathTracer::SetVolume(uint32_t index, const Volume& volume) {
    if (index >= m_Volumes.size()) throw std::runtime_error("SetVolume: index out of range");
    m_Volumes[index] = volume; UploadVolumes();
}
void PathTracer::AddDensityDataToVolume(uint32_t volumeIndex, uint32_t dx, uint32_t dy, uint32_t dz, const float* density) {
    if (volumeIndex >= m_Volumes.size()) throw std::runtime_error("AddDensityDataToVolume: index out of range");
    if (!m_Ctx) throw std::runtime_error("AddDensityDataToVolume before SetScene");
    int idx = vpt_add_density_grid(m_Ctx, dx, dy, dz, density);
    if (idx < 0) Check(idx, "vpt_add_density_grid");
    m_Volumes[volumeIndex].DensityDataIndex = idx;
    UploadVolumes();
}
void PathTracer::RemoveDensityDataFromVolume(uint32_t volumeIndex) {  // the grid itself stays allocated until the context goes
    if (volumeIndex >= m_Volumes.size()) throw std::runtime_error("RemoveDensityDataFromVolume: index out of range");
    m_Volumes[volumeIndex].DensityDataIndex = -1;
    UploadVolumes();
}
void PathTracer::SetPhaseFunction(PhaseFunction phaseFunction) { m_PhaseFunction = phaseFunction; UploadVolumes(); }

void PathTracer::UploadScene() {
    std::vector<vpt_mesh> meshes; std::vector<vpt_instance> inst; std::vector<vpt_texture> tex;
    for (const auto& m : m_Scene.Meshes) meshes.push_back({m.Vertices.data(), (uint32_t)m.Vertices.size(), m.Indices.data(), (uint32_t)m.Indices.size()});
    for (const auto& i : m_Scene.MeshInstances) { vpt_instance v; v.mesh_index = i.MeshIndex; v.material_index = i.MaterialIndex; std::memcpy(v.transform, i.Transform.m, 64); inst.push_back(v); }
    for (const auto& t : m_Scene.Textures) tex.push_back({t.Width, t.Height, t.Channels, t.Data.data()});
    vpt_scene_desc d{};
    d.meshes = meshes.data(); d.mesh_count = (uint32_t)meshes.size();
    d.materials = m_Materials.data(); d.material_count = (uint32_t)m_Materials.size();
    d.instances = inst.data(); d.instance_count = (uint32_t)inst.size();
    d.textures = tex.data(); d.texture_count = (uint32_t)tex.size();
    d.env_rgba = m_Env.data(); d.env_width = m_EnvW; d.env_height = m_EnvH;
    d.lut_reflection = m_LutR.data(); d.lut_refraction_outside = m_LutO.data(); d.lut_refraction_inside = m_LutI.data();
    Check(vpt_set_scene(m_Ctx, &d), "vpt_set_scene");
}

bool PathTracer::PathTrace(uint32_t dispatches) {
    if (!m_Ctx) throw std::runtime_error("PathTrace before SetScene");
    int done = 0;
    Check(vpt_render(m_Ctx, dispatches, &done), "vpt_render");
    vpt_stats st; Check(vpt_get_stats(m_Ctx, &st), "vpt_get_stats");
    m_DispatchCount = st.dispatches;
    m_SamplesAccumulated = (uint32_t)st.frames * m_Params.samples_per_frame;  // PathTracer.cpp:151-153
    return done != 0;
}

bool PathTracer::PathTraceAsync(uint32_t dispatches, uint64_t* ticket) {
    if (!m_Ctx) throw std::runtime_error("PathTrace before SetScene");
    int done = 0;
    Check(vpt_render_async(m_Ctx, dispatches, &done, ticket), "vpt_render_async");
    // the counters advance at record time, as PathTrace's do (PathTracer.cpp:141-153); vpt_get_stats would wait for the device.  Unconditionally:
    // a call can make progress AND report done (2 dispatches left, 4 asked for: 2 run, then max_samples is reached), and the clamp makes the
    // update a no-op when nothing ran
    {
        const uint64_t s2 = (uint64_t)m_Params.screen_chunk_count * m_Params.screen_chunk_count;
        const uint64_t frames_needed = ((uint64_t)m_Params.max_samples + m_Params.samples_per_frame - 1) / m_Params.samples_per_frame;
        m_DispatchCount = std::min<uint64_t>(m_DispatchCount + dispatches, frames_needed * s2);
        m_SamplesAccumulated = (uint32_t)(m_DispatchCount / s2) * m_Params.samples_per_frame;
    }
    return done != 0;
}
void PathTracer::Wait(uint64_t ticket) {
    if (m_Ctx) Check(vpt_wait(m_Ctx, ticket), "vpt_wait");
}

void PathTracer::ResizeImage(uint32_t width, uint32_t height) {
    m_Width = width; m_Height = height;
    if (m_Ctx) Check(vpt_resize(m_Ctx, width, height), "vpt_resize");
    ResetPathTracing();
}

const std::vector<float>& PathTracer::GetOutputImage() {
    m_Output.resize((size_t)m_Width * m_Height * 4);
    Check(vpt_get_radiance(m_Ctx, m_Output.data()), "vpt_get_radiance");
    return m_Output;
}

void PathTracer::SetMaterial(uint32_t index, const Material& material) {  // PathTracer.cpp:712-810
    if (index >= m_Materials.size()) throw std::runtime_error("SetMaterial: index out of range");
    Check(vpt_set_material(m_Ctx, index, &material), "vpt_set_material");
    m_Materials[index] = material;
    ResetPathTracing();
}

void PathTracer::SetUseRayQueries(bool value) { SetFlag(VPT_FLAG_RAY_QUERIES, value); }   // PathTracer.cpp:1086-1096 (false: RTCommon.slang:64-84, include/vpt.h VPT_FLAG_RAY_QUERIES)
void PathTracer::SetCameraViewInverse(const Mat4& view) { m_CameraViewInverse = view; if (m_Ctx) Check(vpt_set_camera(m_Ctx, m_CameraViewInverse.m, m_CameraProjectionInverse.m), "vpt_set_camera"); ResetPathTracing(); }
void PathTracer::SetCameraProjectionInverse(const Mat4& p) { m_CameraProjectionInverse = p; if (m_Ctx) Check(vpt_set_camera(m_Ctx, m_CameraViewInverse.m, m_CameraProjectionInverse.m), "vpt_set_camera"); ResetPathTracing(); }
void PathTracer::SetFlag(uint32_t bit, bool value) { m_Params.flags = value ? (m_Params.flags | bit) : (m_Params.flags & ~bit); Push(true); }
void PathTracer::Push(bool resets) {
    // vpt_set_params keeps the accumulated image when only max_samples changes (SetMaxSamplesAccumulated does not reset
    // upstream, PathTracer.cpp:1003-1006) and resets it for every other field
    if (m_Ctx) Check(vpt_set_params(m_Ctx, &m_Params), "vpt_set_params");
    if (resets) { m_SamplesAccumulated = 0; m_DispatchCount = 0; }
}
void PathTracer::SetEnvironmentMap(const std::vector<float>& rgba, uint32_t width, uint32_t height) {
    if (rgba.size() != (size_t)width * height * 4 || width == 0 || height == 0) throw std::runtime_error("SetEnvironmentMap: bad size");
    m_Env = rgba; m_EnvW = width; m_EnvH = height;
    if (m_Ctx) { UploadScene(); Check(vpt_set_camera(m_Ctx, m_CameraViewInverse.m, m_CameraProjectionInverse.m), "vpt_set_camera"); }
    ResetPathTracing();
}
void PathTracer::SetEnvMapFilepath(const std::string& filePath) {  // PathTracer.cpp:1137-1164 (ImportTexture of an .hdr)
    std::vector<float> rgba; uint32_t w = 0, h = 0; std::string err;
    if (!LoadHDR(filePath, rgba, w, h, err)) throw std::runtime_error(err);
    m_EnvMapFilepath = filePath;
    SetEnvironmentMap(rgba, w, h);
}
void PathTracer::ResetPathTracing() { m_SamplesAccumulated = 0; m_DispatchCount = 0; if (m_Ctx) vpt_reset(m_Ctx); }

}  // namespace vpthost
