"""ctypes mirror of include/vpt.h (struct layouts and prototypes). Plumbing only."""
import ctypes as C

VPT_OK = 0
ERR_NAMES = {
    -1: "VPT_ERR_INVALID_ARGUMENT", -2: "VPT_ERR_NO_DEVICE", -3: "VPT_ERR_OUT_OF_MEMORY", -4: "VPT_ERR_NO_SCENE",
    -5: "VPT_ERR_DEVICE", -6: "VPT_ERR_UNSUPPORTED", -7: "VPT_ERR_LIMIT",
}

FLAG_SKY_MIS = 1 << 0
FLAG_MESH_MIS = 1 << 1
FLAG_LOCAL_HITS = 1 << 8
FLAG_SHOW_ENV_DIRECTLY = 1 << 2
FLAG_GEOMETRY_NORMALS = 1 << 3
FLAG_ENERGY_COMPENSATION = 1 << 4
FLAG_FURNACE = 1 << 5
FLAG_RAY_QUERIES = 1 << 6
FLAG_TONEMAP_LINEAR_BLOOM_TAP = 1 << 7
FLAGS_DEFAULT = (FLAG_SKY_MIS | FLAG_MESH_MIS | FLAG_SHOW_ENV_DIRECTLY | FLAG_ENERGY_COMPENSATION |
                 FLAG_RAY_QUERIES | FLAG_TONEMAP_LINEAR_BLOOM_TAP)

KERNEL_NAMES = ["primary", "extend", "shade", "connect", "bounce", "resolve", "bloom", "tonemap", "shadow", "join"]
KERNEL_COUNT = 10
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_STAGED = 0, 1, 2
PIPELINE_STAGED_R1, PIPELINE_STAGED_SORTED, PIPELINE_WHOLE = 3, 4, 5
LAB_LANES, LAB_LANE_GRID, LAB_TAIL_GRID, LAB_WHOLE_FRAMES, LAB_WHOLE_SCHED = 1, 2, 3, 4, 5
ASYNC_MAX_BOUNCES = 32


class Material(C.Structure):
    _fields_ = [
        ("base_color", C.c_float * 3), ("emissive_color", C.c_float * 3), ("specular_color", C.c_float * 3),
        ("medium_color", C.c_float * 3), ("medium_emissive_color", C.c_float * 3),
        ("metallic", C.c_float), ("roughness", C.c_float), ("ior", C.c_float), ("transmission", C.c_float),
        ("anisotropy", C.c_float), ("anisotropy_rotation", C.c_float), ("medium_density", C.c_float),
        ("medium_anisotropy", C.c_float),
        ("base_color_texture", C.c_uint32), ("normal_texture", C.c_uint32), ("roughness_texture", C.c_uint32),
        ("metallic_texture", C.c_uint32), ("emissive_texture", C.c_uint32),
    ]


assert C.sizeof(Material) == 112


class Volume(C.Structure):  # vpt_volume
    _fields_ = [
        ("corner_min", C.c_float * 3), ("corner_max", C.c_float * 3), ("color", C.c_float * 3), ("emissive_color", C.c_float * 3),
        ("density", C.c_float), ("anisotropy", C.c_float), ("alpha", C.c_float), ("droplet_size", C.c_float),
        ("density_data_index", C.c_int32), ("approximated_scattering", C.c_int32), ("approximated_scattering_falloff", C.c_float),
        ("grid_sharpness", C.c_float), ("has_temperature_data", C.c_int32), ("use_blackbody", C.c_int32), ("temperature_color", C.c_float * 3),
        ("temperature_gamma", C.c_float), ("temperature_scale", C.c_float), ("emissive_color_gamma", C.c_float),
        ("kelvin_min", C.c_int32), ("kelvin_max", C.c_int32),
    ]


assert C.sizeof(Volume) == 120


class Atmosphere(C.Structure):  # vpt_atmosphere
    _fields_ = [
        ("planet_position", C.c_float * 3), ("planet_radius", C.c_float), ("atmosphere_height", C.c_float),
        ("rayleigh_density_falloff", C.c_float), ("mie_density_falloff", C.c_float), ("ozone_density_falloff", C.c_float),
        ("ozone_peak", C.c_float),
        ("rayleigh_multiplier", C.c_float * 3), ("mie_multiplier", C.c_float * 3), ("ozone_multiplier", C.c_float * 3),
        ("sun_color", C.c_float * 3),
    ]


assert C.sizeof(Atmosphere) == 84


def atmosphere(**kw):
    """PathTracer.h:222-232 defaults (metres; the planet centre sits 1 km + one radius along +Y, i.e. below a Y-down world)."""
    a = Atmosphere()
    d = dict(planet_position=(0.0, 6360e3 + 1000.0, 0.0), planet_radius=6360e3, atmosphere_height=100e3, rayleigh_density_falloff=8000.0,
             mie_density_falloff=1200.0, ozone_density_falloff=5000.0, ozone_peak=22000.0, rayleigh_multiplier=(1, 1, 1), mie_multiplier=(1, 1, 1),
             ozone_multiplier=(1, 1, 1), sun_color=(1.0, 0.956, 0.88))
    d.update(kw)
    for k, v in d.items():
        if isinstance(v, (tuple, list)):
            getattr(a, k)[:] = v
        else:
            setattr(a, k, v)
    return a
PHASE_HENYEY_GREENSTEIN, PHASE_DRAINE, PHASE_HENYEY_GREENSTEIN_PLUS_DRAINE = 0, 1, 2


def volume(corner_min=(-1, -1, -1), corner_max=(1, 1, 1), color=(0.8, 0.8, 0.8), emissive_color=(0, 0, 0), density=1.0, anisotropy=0.0,
           alpha=1.0, droplet_size=20.0, approximated_scattering=0, approximated_scattering_falloff=0.8, density_data_index=-1, grid_sharpness=1.0,
           has_temperature_data=0, use_blackbody=1, temperature_color=(1.0, 0.5, 0.0), temperature_gamma=1.0, temperature_scale=1.0,
           emissive_color_gamma=1.0, kelvin_min=500, kelvin_max=8000):
    """PathTracer::Volume defaults (PathTracer.h:36-74); corners are world space (Position + Corner * Scale applied)."""
    v = Volume()
    v.corner_min[:] = corner_min; v.corner_max[:] = corner_max; v.color[:] = color; v.emissive_color[:] = emissive_color
    v.density, v.anisotropy, v.alpha, v.droplet_size = density, anisotropy, alpha, droplet_size
    v.density_data_index = density_data_index
    v.grid_sharpness = grid_sharpness
    v.has_temperature_data, v.use_blackbody = has_temperature_data, use_blackbody
    v.temperature_color[:] = temperature_color
    v.temperature_gamma, v.temperature_scale, v.emissive_color_gamma = temperature_gamma, temperature_scale, emissive_color_gamma
    v.kelvin_min, v.kelvin_max = kelvin_min, kelvin_max
    v.approximated_scattering, v.approximated_scattering_falloff = approximated_scattering, approximated_scattering_falloff
    return v


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p),
                ("index_count", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("mesh_index", C.c_uint32), ("material_index", C.c_uint32), ("transform", C.c_float * 16)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("data", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_uint32),
        ("materials", C.POINTER(Material)), ("material_count", C.c_uint32),
        ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint32),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
        ("env_rgba", C.c_void_p), ("env_width", C.c_uint32), ("env_height", C.c_uint32),
        ("lut_reflection", C.c_void_p), ("lut_refraction_outside", C.c_void_p), ("lut_refraction_inside", C.c_void_p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("samples_per_frame", C.c_uint32), ("max_samples", C.c_uint32), ("max_depth", C.c_uint32),
        ("max_luminance", C.c_float), ("focus_distance", C.c_float), ("dof_strength", C.c_float),
        ("sky_azimuth", C.c_float), ("sky_altitude", C.c_float), ("sky_intensity", C.c_float),
        ("screen_chunk_count", C.c_uint32), ("emissive_pdf_bias", C.c_float), ("flags", C.c_uint32),
        ("base_seed", C.c_uint32),
    ]


def default_params(**kw):
    """PathTracer.h:197-233 defaults."""
    p = Params(1, 5000, 200, 500.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1, 0.0, FLAGS_DEFAULT, 1)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class PostParams(C.Structure):
    _fields_ = [("exposure", C.c_float), ("gamma", C.c_float), ("bloom_threshold", C.c_float),
                ("bloom_strength", C.c_float), ("mip_count", C.c_uint32), ("falloff_range", C.c_float), ("schedule", C.c_uint32)]


def default_post_params(**kw):
    """PostProcessor.h:8-21 defaults."""
    p = PostParams(1.0, 2.2, 2.0, 1.0, 10, 5.0, 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("width", C.c_uint32), ("height", C.c_uint32), ("shard_rank", C.c_uint32),
                ("shard_count", C.c_uint32), ("frames_in_flight", C.c_uint32), ("profile", C.c_uint32),
                ("count_traversal", C.c_uint32), ("pipeline", C.c_uint32), ("build_flags", C.c_uint32), ("resident_frames", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("frames", C.c_uint64), ("dispatches", C.c_uint64), ("closest_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64), ("connect_paths", C.c_uint64), ("primary_hits", C.c_uint64), ("primary_survivors", C.c_uint64), ("primary_shadow_rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
        ("shadow_nodes_visited", C.c_uint64), ("shadow_tris_tested", C.c_uint64),
        ("kernel_launches", C.c_uint64 * KERNEL_COUNT), ("kernel_ms", C.c_double * KERNEL_COUNT),
        ("total_vertex_count", C.c_uint64), ("total_index_count", C.c_uint64),
        ("bvh_nodes", C.c_uint32), ("bvh_triangles", C.c_uint32), ("bvh_node_bytes", C.c_uint32),
        ("bvh_tri_bytes", C.c_uint32), ("emissive_mesh_count", C.c_uint32), ("emissive_triangle_count", C.c_uint32),
        ("frames_in_flight", C.c_uint32), ("shard_pixels", C.c_uint32), ("bvh8_nodes", C.c_uint32), ("build_flags", C.c_uint32),
        ("frames_allocated", C.c_uint32), ("resident_frames", C.c_uint32), ("reserved0", C.c_uint32), ("graph_launches", C.c_uint32), ("stack_spills", C.c_uint64 * 2),
        ("set_scene_ms", C.c_double), ("bvh_build_ms", C.c_double),
        ("finish_paths", C.c_uint64), ("finish_closest_rays", C.c_uint64), ("finish_shadow_rays", C.c_uint64),
    ]


class CommInfo(C.Structure):
    _fields_ = [("rccl_version_runtime", C.c_int32), ("rccl_version_compiled", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32), ("library_path", C.c_char * 232)]


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("tmin", C.c_float), ("direction", C.c_float * 3), ("tmax", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("primitive", C.c_uint32),
                ("instance", C.c_uint32)]


# every symbol include/vpt.h declares, with (restype, argtypes)
PROTOTYPES = {
    "vpt_create": (C.c_void_p, [C.POINTER(Config), C.POINTER(C.c_int)]),
    "vpt_destroy": (None, [C.c_void_p]),
    "vpt_last_error": (C.c_char_p, [C.c_void_p]),
    "vpt_set_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "vpt_set_material": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Material)]),
    "vpt_get_material": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Material)]),
    "vpt_set_camera": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vpt_default_atmosphere": (None, [C.POINTER(Atmosphere)]),
    "vpt_set_atmosphere": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vpt_add_density_grid": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "vpt_clear_density_grids": (C.c_int, [C.c_void_p]),
    "vpt_set_volumes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "vpt_set_phase_function": (C.c_int, [C.c_void_p, C.c_uint32]),
    "vpt_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "vpt_default_params": (None, [C.POINTER(Params)]),
    "vpt_default_post_params": (None, [C.POINTER(PostParams)]),
    "vpt_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "vpt_reset": (C.c_int, [C.c_void_p]),
    "vpt_render": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]),
    "vpt_render_async": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vpt_postprocess_device": (C.c_int, [C.c_void_p, C.POINTER(PostParams), C.c_void_p, C.POINTER(C.c_uint64)]),
    "vpt_wait": (C.c_int, [C.c_void_p, C.c_uint64]),
    "vpt_output_device": (C.c_void_p, [C.c_void_p]),
    "vpt_get_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vpt_get_radiance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vpt_get_radiance_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vpt_set_radiance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "vpt_shard_floats": (C.c_size_t, [C.c_void_p]),
    "vpt_get_shard_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vpt_assemble_shards": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "vpt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vpt_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vpt_comm_gather_shards": (C.c_int, [C.c_void_p, C.c_int]),
    "vpt_comm_get_info": (C.c_int, [C.c_void_p, C.POINTER(CommInfo)]),
    "vpt_device_identity": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32]),
    "vpt_comm_destroy": (C.c_int, [C.c_void_p]),
    "vpt_multi_gather_shards": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32]),
    "vpt_postprocess": (C.c_int, [C.c_void_p, C.POINTER(PostParams), C.c_void_p, C.c_void_p]),
    "vpt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "vpt_reset_stats": (C.c_int, [C.c_void_p]),
    "vpt_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vpt_lab_set": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "vpt_lab_set_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "vpt_lab_trace": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "vpt_lut_calculate": (C.c_int, [C.c_int] + [C.c_uint32] * 6 + [C.c_void_p]),
}


COMM_ID_BYTES = 128


LAB_ENTRY_POINTS = ("vpt_lab_set", "vpt_lab_set_rays", "vpt_lab_trace")   # include/vpt_lab.h: exported by the laboratory build only


def bind(lib):
    """Sets the prototypes of every entry point of include/vpt.h (all must be present) and of include/vpt_lab.h (present in
    libvpt_hip_lab.so only); lib.has_lab says which library this is."""
    lib.has_lab = True
    for name, (res, args) in PROTOTYPES.items():
        if name in LAB_ENTRY_POINTS and not hasattr(lib, name):
            lib.has_lab = False
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
