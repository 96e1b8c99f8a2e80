// SceneLoader.h — stands in for VulkanHelper::AssetImporter (absent submodule; reference call sites
// PathTracer.cpp:166-167, 814-815, 1139-1140): a minimal glTF 2.0 reader producing exactly the structs
// PathTracer::SetScene consumes (Scene{Meshes, Materials, MeshInstances, Cameras}).
// Same conventions as vulkan-path-tracer_amd/scenes.py:load_gltf (Y-down world, winding swapped).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vpt.h"
#include "Math.h"

namespace vpthost {

struct TextureAsset {  // VulkanHelper::TextureAsset (PathTracer.cpp:815-836)
    uint32_t Width = 0, Height = 0, Channels = 4;
    std::vector<uint8_t> Data;
};
struct MeshAsset {
    std::vector<vpt_vertex> Vertices;  // == LoadedMeshVertex, 32 B
    std::vector<uint32_t> Indices;
};
struct MeshInstance {
    uint32_t MeshIndex = 0, MaterialIndex = 0;
    Mat4 Transform;
};
struct CameraAsset {
    float AspectRatio = 16.0f / 9.0f, FOV = 45.0f;
    Mat4 ViewMatrix;
};
struct SceneAsset {
    std::vector<MeshAsset> Meshes;
    std::vector<vpt_material> Materials;
    std::vector<std::string> MaterialNames;
    std::vector<MeshInstance> MeshInstances;
    std::vector<CameraAsset> Cameras;
    std::vector<TextureAsset> Textures;  // index 0..4: the default textures of LoadDefaultTexture, then loaded ones
};

// Returns false and fills `error` on failure (the reference aborts via VH_ASSERT, PathTracer.cpp:168).
bool ImportScene(const std::string& gltfPath, SceneAsset& out, std::string& error);
// LDR textures as stbi_load(path, 4) delivers them to LoadTexture (PathTracer.cpp:812-836): RGBA8, rows top to bottom (ImageCodec.cpp).
// PNG: every colour type and bit depth of the format, palette, tRNS, Adam7.  JPEG: baseline and progressive Huffman, 8 bit,
// 1 or 3 components, any sampling factors, restart intervals; stb_image's IDCT / chroma upsampling / YCbCr arithmetic.
bool LoadPNG(const std::string& path, TextureAsset& out, std::string& error);
bool LoadImage(const std::string& path, TextureAsset& out, std::string& error);   // PNG or JPEG, decided by content
bool DecodePNG(const std::string& bytes, const std::string& name, TextureAsset& out, std::string& error);
bool DecodeJPEG(const std::string& bytes, const std::string& name, TextureAsset& out, std::string& error);
bool DecodeImage(const std::string& bytes, const std::string& name, TextureAsset& out, std::string& error);
// Radiance .hdr (RGBE; flat or new-style RLE scanlines; "-Y h +X w" orientation) -> RGBA32F rows top to bottom, A = 1:
// what ImportTexture hands LoadEnvironmentMap for the default env map (PathTracer.cpp:1137-1164, PathTracer.h:208).
// value = mantissa * 2^(e - 136), 0 when e == 0 (the stb_image convention the reference's importer follows).
bool LoadHDR(const std::string& path, std::vector<float>& rgba, uint32_t& width, uint32_t& height, std::string& error);
// 8-bit RGBA PNG (Editor::SaveToFile -> stbi_write_png, Editor.cpp:815-843): filter 0 rows, one zlib stream.
bool SavePNG(const std::string& path, const uint8_t* rgba, uint32_t width, uint32_t height, std::string& error);
// Assets/LookupTables as one raw fp32 file (reflection 64x64x32, refraction outside/inside 128x128x32).
bool LoadLookupTables(const std::string& path, std::vector<float>& reflection, std::vector<float>& outside, std::vector<float>& inside, std::string& error);

}  // namespace vpthost
