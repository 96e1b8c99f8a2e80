// vpt_render — headless driver in place of Application / Editor (Application.cpp:81-99, Editor.cpp:81-143):
// load a glTF scene, run PathTrace() until the sample budget is spent, post-process, write the images.
//   vpt_render --scene S.gltf --luts lookup_tables.bin [--size WxH] [--spp N] [--depth D] [--seed K] [--split S]
//              [--env-constant r,g,b] [--radiance out.f32] [--camera out.f32] [--ppm out.ppm] [--info] [--dump-scene out.bin]
//              [--env-hdr sky.hdr] [--png out.png] [--atmosphere] [--sun altitude,azimuth]
//              [--volume minx,miny,minz,maxx,maxy,maxz,density,g,r,g,b]... [--phase hg|draine|hg+draine]
//              [--async]   one PathTraceAsync + PostProcessAsync per frame with a one-frame fence lag (the reference's Editor loop) instead of blocking batches
//              [--async-step N]   dispatches per PathTraceAsync call (default 1)
//              [--no-ray-queries]   SetUseRayQueries(false): the TraceRay forms of the shadow / distance queries (RTCommon.slang:64-84)
//              [--gpus N [--devices d0,d1,...]]   rows y % N == k rendered on device k (one host thread each), one gather of
//                                                 the shards into device 0 over xGMI, post-process there; the image is
//                                                 bit-identical for every N (a device may be listed more than once)
//   vpt_render --make-lut reflect|refract-above|refract-below --lut-samples N [--lut-size XxYxZ] [--lut-time-seed T] --lut-out table.bin
//              (Application.cpp:38-77: the three tables the reference regenerates with 10'000'000 samples)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>

#include "FlyCamera.h"
#include "LookupTableCalculator.h"
#include "PathTracer.h"
#include "PostProcessor.h"

using namespace vpthost;

static void write_file(const std::string& path, const void* data, size_t bytes) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f.write((const char*)data, (std::streamsize)bytes);
}

int main(int argc, char** argv) {
    std::string scene, luts, radiance, camera, ppm, png, envHdr, dumpEnv, pngTest, decodeImage, dumpImage, dump, makeLut, lutOut;
    std::vector<PathTracer::Volume> volumes; int phase = 0; bool atmosphere = false; float sunAlt = 0.0f, sunAz = 0.0f;
    uint32_t lutSamples = 10000000u, lutTime = 0; UVec3 lutSize{0, 0, 0};
    uint32_t w = 0, h = 0, spp = 16, depth = 8, seed = 1, split = 1, gpus = 1; std::vector<int> devices;
    bool async = false, rayQueries = true; uint32_t asyncStep = 1;
    bool info = false, selftest = false; float env[3] = {0, 0, 0}; bool haveEnv = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--luts") luts = next();
        else if (a == "--size") { std::string s = next(); if (sscanf(s.c_str(), "%ux%u", &w, &h) != 2) { fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--spp") spp = (uint32_t)atoi(next().c_str());
        else if (a == "--async") async = true;
        else if (a == "--no-ray-queries") rayQueries = false;
        else if (a == "--async-step") { int v = atoi(next().c_str()); asyncStep = v > 1 ? (uint32_t)v : 1u; }
        else if (a == "--depth") depth = (uint32_t)atoi(next().c_str());
        else if (a == "--seed") seed = (uint32_t)strtoul(next().c_str(), nullptr, 10);
        else if (a == "--split") split = (uint32_t)atoi(next().c_str());
        else if (a == "--gpus") { gpus = (uint32_t)atoi(next().c_str()); if (gpus == 0 || gpus > 64) { fprintf(stderr, "--gpus 1..64\n"); return 2; } }
        else if (a == "--devices") { std::string v = next(); size_t p0 = 0; while (p0 <= v.size()) { size_t q = v.find(',', p0); if (q == std::string::npos) q = v.size(); devices.push_back(atoi(v.substr(p0, q - p0).c_str())); p0 = q + 1; } }
        else if (a == "--env-constant") { std::string s = next(); if (sscanf(s.c_str(), "%f,%f,%f", &env[0], &env[1], &env[2]) != 3) return 2; haveEnv = true; }
        else if (a == "--radiance") radiance = next();
        else if (a == "--camera") camera = next();
        else if (a == "--ppm") ppm = next();
        else if (a == "--png") png = next();                 // Editor::SaveToFile
        else if (a == "--env-hdr") envHdr = next();          // SetEnvMapFilepath
        else if (a == "--dump-env") dumpEnv = next();        // with --env-hdr: decoded RGBA32F (no device needed)
        else if (a == "--decode-image") decodeImage = next(); // with --dump-image: a PNG / JPEG texture as the importer decodes it, raw RGBA8 (no device needed)
        else if (a == "--dump-image") dumpImage = next();
        else if (a == "--png-roundtrip") pngTest = next();   // with --dump-env unused: writes a test pattern PNG and reads it back (no device needed)
        else if (a == "--info") info = true;
        else if (a == "--selftest") selftest = true;
        else if (a == "--dump-scene") { dump = next(); info = true; }
        else if (a == "--volume") {
            PathTracer::Volume v; float f[11]; std::string s = next();
            if (sscanf(s.c_str(), "%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5], &f[6], &f[7], &f[8], &f[9], &f[10]) != 11) { fprintf(stderr, "--volume needs 11 numbers\n"); return 2; }
            v.CornerMin = Vec3(f[0], f[1], f[2]); v.CornerMax = Vec3(f[3], f[4], f[5]); v.Density = f[6]; v.Anisotropy = f[7]; v.Color = Vec3(f[8], f[9], f[10]);
            volumes.push_back(v);
        }
        else if (a == "--phase") { std::string s = next(); phase = s == "hg" ? 0 : s == "draine" ? 1 : s == "hg+draine" ? 2 : -1; if (phase < 0) { fprintf(stderr, "--phase hg|draine|hg+draine\n"); return 2; } }
        else if (a == "--atmosphere") atmosphere = true;
        else if (a == "--sun") { std::string s = next(); if (sscanf(s.c_str(), "%f,%f", &sunAlt, &sunAz) != 2) { fprintf(stderr, "--sun altitude,azimuth\n"); return 2; } }
        else if (a == "--make-lut") makeLut = next();
        else if (a == "--lut-samples") lutSamples = (uint32_t)strtoul(next().c_str(), nullptr, 10);
        else if (a == "--lut-time-seed") lutTime = (uint32_t)strtoul(next().c_str(), nullptr, 10);
        else if (a == "--lut-out") lutOut = next();
        else if (a == "--lut-size") { std::string s = next(); if (sscanf(s.c_str(), "%ux%ux%u", &lutSize.x, &lutSize.y, &lutSize.z) != 3) { fprintf(stderr, "--lut-size XxYxZ\n"); return 2; } }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (selftest) {  // host-side arithmetic only (no device): FlyCamera <-> matrices, Mat4 inverse
        Mat4 view = lookAt(Vec3(1.0f, -2.0f, 6.0f), Vec3(0.2f, -0.5f, 0.0f), Vec3(0.0f, 1.0f, 0.0f));
        Mat4 proj = perspective(radians(45.0f), 16.0f / 9.0f, 0.1f, 1000.0f);
        FlyCamera cam(view, proj);
        Mat4 v2 = cam.GetViewMatrix(), p2 = cam.GetProjectionMatrix(), id = multiply(view, inverse(view));
        double ev = 0, ep = 0, ei = 0;
        for (int i = 0; i < 16; i++) { ev = std::fmax(ev, std::fabs(v2.m[i] - view.m[i])); ep = std::fmax(ep, std::fabs(p2.m[i] - proj.m[i])); ei = std::fmax(ei, std::fabs(id.m[i] - (i % 5 == 0 ? 1.0f : 0.0f))); }
        cam.ProcessKeyboard(FlyCamera::Direction::UP, 1.0f);  // UP moves against m_Up (FlyCamera.cpp:47-49)
        // a moved PathTracer carries ALL of its state (volumes, phase function, atmosphere, env path, shard identity)
        PathTracer a = PathTracer::New(0, 1, 4), b = PathTracer::New(0);
        PathTracer::Volume vol; vol.Density = 3.0f;
        a.AddVolume(vol); a.AddVolume(vol); a.SetPhaseFunction(PathTracer::PhaseFunction::DRAINE); a.SetPlanetRadius(1234.0f); a.SetEnableAtmosphere(true);
        a.SetMaxDepth(7);
        b = std::move(a);
        PathTracer c2(std::move(b));
        const bool moveOk = c2.GetVolumesCount() == 2 && c2.GetVolumes()[1].Density == 3.0f && c2.GetPhaseFunction() == PathTracer::PhaseFunction::DRAINE &&
                            c2.IsAtmosphereEnabled() && c2.GetPlanetRadius() == 1234.0f && c2.GetMaxDepth() == 7 && c2.GetShardRank() == 1 && c2.GetShardCount() == 4 &&
                            b.GetVolumesCount() == 0 && !b.IsAtmosphereEnabled();
        printf("{\"view_err\": %.3g, \"proj_err\": %.3g, \"inverse_err\": %.3g, \"fov\": %.4f, \"aspect\": %.5f, \"up_dy\": %.4f, \"move_keeps_state\": %s}\n", ev, ep, ei, cam.GetFov(), cam.GetAspectRatio(),
               cam.GetPosition().y - (-2.0f), moveOk ? "true" : "false");
        return (ev < 1e-4 && ep < 1e-5 && ei < 1e-5 && moveOk) ? 0 : 1;
    }
    if (!dumpEnv.empty() || !pngTest.empty() || !decodeImage.empty()) {  // host-side file formats only
        try {
            std::string err;
            if (!decodeImage.empty()) {
                TextureAsset t;
                if (!LoadImage(decodeImage, t, err)) throw std::runtime_error(err);
                if (!dumpImage.empty()) write_file(dumpImage, t.Data.data(), t.Data.size());
                printf("{\"width\": %u, \"height\": %u, \"channels\": %u}\n", t.Width, t.Height, t.Channels);
            }
            if (!dumpEnv.empty()) {
                std::vector<float> rgba; uint32_t ew = 0, eh = 0;
                if (!LoadHDR(envHdr, rgba, ew, eh, err)) throw std::runtime_error(err);
                write_file(dumpEnv, rgba.data(), rgba.size() * 4);
                printf("{\"width\": %u, \"height\": %u}\n", ew, eh);
            }
            if (!pngTest.empty()) {
                const uint32_t pw = 37, ph = 21;
                std::vector<uint8_t> px((size_t)pw * ph * 4);
                for (size_t i = 0; i < px.size(); i++) px[i] = (uint8_t)((i * 2654435761u) >> 13);
                if (!SavePNG(pngTest, px.data(), pw, ph, err)) throw std::runtime_error(err);
                TextureAsset back;
                if (!LoadPNG(pngTest, back, err)) throw std::runtime_error(err);
                if (back.Width != pw || back.Height != ph || back.Data != px) throw std::runtime_error("PNG round trip differs");
                printf("{\"png_roundtrip\": true}\n");
            }
            return 0;
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (!makeLut.empty()) {
        try {
            const bool reflect = makeLut == "reflect";
            if (!reflect && makeLut != "refract-above" && makeLut != "refract-below") throw std::runtime_error("--make-lut reflect|refract-above|refract-below");
            if (lutOut.empty()) throw std::runtime_error("--lut-out is required");
            std::vector<ShaderDefine> defs;
            if (!reflect) defs.push_back({makeLut == "refract-above" ? "ABOVE_SURFACE" : "BELOW_SURFACE", ""});
            LookupTableCalculator calc = LookupTableCalculator::New(0, reflect ? "LookupReflect.slang" : "LookupRefract.slang", defs);
            calc.SetTimeSeed(lutTime);
            if (lutSize.x == 0) lutSize = reflect ? UVec3{64, 64, 32} : UVec3{128, 128, 32};  // Application.cpp:41,54,67
            auto t0 = std::chrono::steady_clock::now();
            std::vector<float> table = calc.CalculateTable(lutSize, lutSamples);
            double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            write_file(lutOut, table.data(), table.size() * 4);
            printf("{\"table\": \"%s\", \"size\": [%u, %u, %u], \"samples_per_cell\": %u, \"seconds\": %.3f, \"gsamples_per_s\": %.2f}\n", makeLut.c_str(), lutSize.x,
                   lutSize.y, lutSize.z, lutSamples, sec, (double)table.size() * (lutSamples / 20u * 20u) / sec * 1e-9);
            return 0;
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (scene.empty()) { fprintf(stderr, "usage: vpt_render --scene S.gltf --luts lookup_tables.bin [--size WxH] [--spp N] [--depth D] ...\n"); return 2; }
    try {
        if (info) {  // scene import only: no device needed
            SceneAsset sc; std::string err;
            if (!ImportScene(scene, sc, err)) throw std::runtime_error(err);
            if (!dump.empty()) {  // flat binary image of the imported scene, for comparison with the Python loader
                std::string b;
                auto put = [&](const void* p, size_t n) { b.append((const char*)p, n); };
                auto u32 = [&](uint32_t v) { put(&v, 4); };
                u32((uint32_t)sc.Meshes.size());
                for (auto& m : sc.Meshes) { u32((uint32_t)m.Vertices.size()); u32((uint32_t)m.Indices.size()); put(m.Vertices.data(), m.Vertices.size() * sizeof(vpt_vertex)); put(m.Indices.data(), m.Indices.size() * 4); }
                u32((uint32_t)sc.Materials.size()); put(sc.Materials.data(), sc.Materials.size() * sizeof(vpt_material));
                u32((uint32_t)sc.MeshInstances.size());
                for (auto& i : sc.MeshInstances) { u32(i.MeshIndex); u32(i.MaterialIndex); put(i.Transform.m, 64); }
                u32((uint32_t)sc.Textures.size());
                for (auto& t : sc.Textures) { u32(t.Width); u32(t.Height); u32(t.Channels); put(t.Data.data(), t.Data.size()); }
                u32((uint32_t)sc.Cameras.size());
                for (auto& c : sc.Cameras) { put(&c.AspectRatio, 4); Mat4 vi = inverse(c.ViewMatrix); put(vi.m, 64); }
                write_file(dump, b.data(), b.size());
            }
            size_t tris = 0, verts = 0;
            for (auto& i : sc.MeshInstances) tris += sc.Meshes[i.MeshIndex].Indices.size() / 3;
            for (auto& m : sc.Meshes) verts += m.Vertices.size();
            printf("{\"meshes\": %zu, \"instances\": %zu, \"materials\": %zu, \"textures\": %zu, \"triangles\": %zu, \"vertices\": %zu, \"cameras\": %zu}\n",
                   sc.Meshes.size(), sc.MeshInstances.size(), sc.Materials.size(), sc.Textures.size(), tris, verts, sc.Cameras.size());
            return 0;
        }
        while (devices.size() < gpus) devices.push_back((int)devices.size());
        std::vector<PathTracer> shards;
        for (uint32_t k = 0; k < gpus; k++) shards.push_back(PathTracer::New(devices[k], k, gpus));
        for (PathTracer& pt : shards) {   // every shard holds a replica of the scene and the same settings
            pt.SetLookupTablePath(luts);
            if (haveEnv) { std::vector<float> e(64 * 32 * 4, 0.0f); for (size_t i = 0; i < 64 * 32; i++) { e[i * 4] = env[0]; e[i * 4 + 1] = env[1]; e[i * 4 + 2] = env[2]; } pt.SetEnvironmentMap(e, 64, 32); }
            if (w && h) pt.ResizeImage(w, h);
            if (!envHdr.empty()) pt.SetEnvMapFilepath(envHdr);
            pt.SetScene(scene);
            if (w && h) {  // the window was resized: Editor.cpp:203-211 rebuilds the projection from the new aspect ratio
                FlyCamera cam(inverse(pt.GetCameraViewInverse()), inverse(pt.GetCameraProjectionInverse()));
                cam.SetAspectRatio((float)w / (float)h); cam.SetNearFar(0.1f, 100.0f);
                pt.SetCameraProjectionInverse(inverse(cam.GetProjectionMatrix()));
            }
            pt.SetMaxDepth(depth); pt.SetSeed(seed); pt.SetSplitScreenCount(split); pt.SetMaxSamplesAccumulated(spp);
            if (!rayQueries) pt.SetUseRayQueries(false);
            if (phase != 0) pt.SetPhaseFunction((PathTracer::PhaseFunction)phase);
            for (const auto& v : volumes) pt.AddVolume(v);
            if (sunAlt != 0.0f || sunAz != 0.0f) { pt.SetSkyAltitude(sunAlt); pt.SetSkyAzimuth(sunAz); }
            if (atmosphere) pt.SetEnableAtmosphere(true);
        }
        PathTracer& pt = shards[0];
        auto t0 = std::chrono::steady_clock::now();
        PostProcessor post = PostProcessor::New();
        post.SetInputImage(pt);
        if (gpus == 1 && async) {
            // Editor::Draw's loop (Editor.cpp:116,129): PathTrace(cmd) once and PostProcess(cmd) every frame, both RECORDED, the host
            // waiting on the fence of the frame before — the image and the RGBA8 output stay on the device until they are asked for
            uint64_t prev = 0;
            while (!pt.PathTraceAsync(asyncStep)) {
                const uint64_t t = post.PostProcessAsync();
                if (prev) pt.Wait(prev);
                prev = t;
            }
            pt.Wait();
        } else if (gpus == 1) {
            while (!pt.PathTrace(64)) {}
        } else {  // one host thread per device; the shards never talk to each other until the gather
            std::vector<std::thread> th; std::vector<std::string> errs(gpus);
            for (uint32_t k = 0; k < gpus; k++)
                th.emplace_back([&, k] { try { while (!shards[k].PathTrace(64)) {} } catch (const std::exception& e) { errs[k] = e.what(); } });
            for (auto& t : th) t.join();
            for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
            std::vector<PathTracer*> ps; for (PathTracer& p : shards) ps.push_back(&p);
            PathTracer::GatherShards(ps, 0);
        }
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const std::vector<float>& img = pt.GetOutputImage();
        post.PostProcess();
        if (!radiance.empty()) write_file(radiance, img.data(), img.size() * 4);
        if (!camera.empty()) { float m[32]; memcpy(m, pt.GetCameraViewInverse().m, 64); memcpy(m + 16, pt.GetCameraProjectionInverse().m, 64); write_file(camera, m, sizeof(m)); }
        if (!ppm.empty()) {
            const std::vector<uint8_t>& o = post.GetOutputImage();
            std::string out = "P6\n" + std::to_string(pt.GetWidth()) + " " + std::to_string(pt.GetHeight()) + "\n255\n";
            for (size_t i = 0; i < (size_t)pt.GetWidth() * pt.GetHeight(); i++) out.append((const char*)&o[i * 4], 3);
            write_file(ppm, out.data(), out.size());
        }
        if (!png.empty()) {
            std::string err;
            if (!SavePNG(png, post.GetOutputImage().data(), pt.GetWidth(), pt.GetHeight(), err)) throw std::runtime_error(err);
        }
        printf("{\"width\": %u, \"height\": %u, \"gpus\": %u, \"samples\": %u, \"seconds\": %.4f, \"msamples_per_s\": %.2f, \"vertices\": %llu, \"indices\": %llu}\n", pt.GetWidth(), pt.GetHeight(),
               gpus, pt.GetSamplesAccumulated(), sec, (double)pt.GetWidth() * pt.GetHeight() * pt.GetSamplesAccumulated() / sec / 1e6,
               (unsigned long long)pt.GetTotalVertexCount(), (unsigned long long)pt.GetTotalIndexCount());
    } catch (const std::exception& e) {
        fprintf(stderr, "vpt_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
