/* vpt_lab.h — the LABORATORY's entry points: measurement hooks on the traversal kernels and on the scheduling of asynchronous frames.
 *
 * NOT part of the drop-in boundary (include/vpt.h) and NOT in the product library: libvpt_hip.so exports none of these symbols.  They exist in
 * libvpt_hip_lab.so, the same sources compiled with -DVPT_LAB=1 (python -m "vulkan-path-tracer_amd._build" --lab; VPT_LAB=1 in the environment makes
 * the Python shim load it), together with every kernel variant that was built, measured against the product kernels and found slower
 * (profiles/REJECTED.md: the baseline traversal loop, the eight-wide tree, stale-entry culling, packed fp32 node arithmetic, ray-slot pools, two
 * rays per lane, round 1's stage kernels = VPT_PIPELINE_STAGED_R1).  tests/tools/trace_lab.py, latency_probe.py, whole_*.py drive them.
 * Images never depend on anything here. */
#ifndef VPT_LAB_H
#define VPT_LAB_H
#include "vpt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement hook on the ray-stream traversal kernels alone (the "trace lab"): keep a ray set resident on the device,
 * then time kernel variants on exactly those rays, optionally visiting them in a caller-given order (a permutation of
 * 0..n-1: e.g. sorted by origin cell and direction octant).  All rays of a set share ray 0's tmin / tmax.  Results are
 * per ray whatever the variant or order: hits_host (optional, n entries; any-hit: t = 1 occluded / -1 clear).
 * best_ms = fastest of `reps` launches (HIP events on the context's stream); visits (optional) = {nodes, triangles}
 * visited, from one extra counting launch. */
#define VPT_TRACE_BASE 0u  /* one ray per lane, 64 rays per wave at a time (round 1's extend / shadow loop) */
#define VPT_TRACE_VOTE 1u  /* persistent lanes, wave-level vote between node / triangle / fetch steps, ray replacement */
#define VPT_TRACE_VOTE8 2u /* the same on an eight-wide tree with octant-ordered children (BVH8 experiment; built on first use) */
#define VPT_TRACE_POOL 3u  /* closest hit only: a wave owns 64-128 ray slots in LDS and every step runs on up to 64 of the slots that want it (lanes are
                            * workers, not owners of a ray; kernels_trace.hip k_trace_pool).  param: low byte = idle slots that trigger a fetch (0: 5/16 of the pool); bits 8-9 = slots per wave /
                            * LDS stack entries 128/10, 96/10, 80/8, 64/8 (3-6 blocks per CU); bit 10 = a node step and a triangle step per iteration, loads
                            * of both in flight together; bits 16-21 = slots at leaves that make such an iteration carry the triangle step (0: 32) */
#define VPT_TRACE_PAIR 4u  /* closest hit only: every lane keeps TWO rays in its registers and serves, in a step of the voted kind, whichever of them wants it
                            * (kernels_trace.hip k_trace_pair).  param: low byte = idle rays (of 128 per wave) that trigger a fetch (0: 48) */
#define VPT_TRACE_VOTE4S 5u /* the vote kernel on the SPLIT-ORDER four-wide tree — hit children visited in the order the node's binary splits give for the
                             * ray's direction octant (three table bits, a three-exchange butterfly) instead of sorted by entry distance (vote.hpp vote_node4s_step; built on first use) */
int vpt_lab_set_rays(vpt_ctx* ctx, const vpt_ray* rays_host, uint32_t n);
/* Measurement hook on the scheduling of pipelined 1-frame batches (vpt_render_async; tests/tools/latency_probe.py): images never depend on it.
 *   VPT_LAB_LANES       lanes consecutive frames are dealt to (1-3; default 3: a frame takes the first lane whose previous frame is resolved)
 *   VPT_LAB_LANE_GRID   divisor of the fused kernel's persistent grid while frames are pipelined (1-3; default 1)
 *   VPT_LAB_TAIL_GRID   divisor of the grid of a 1-frame batch's bounces >= 2, whose queues hold a fraction of the frame's paths (1-3; default 3)
 *   VPT_LAB_WHOLE_FRAMES  VPT_PIPELINE_AUTO runs batches of at most this many frames as one whole-path launch where VPT_PIPELINE_WHOLE applies
 *                         (0: never — the per-bounce kernels; 0xffff: no bound, the default)
 *   VPT_LAB_WHOLE_SCHED   how a whole-path launch deals its tiles of 64 samples: low 4 bits = tiles per atomic (1-15), bits 4-5 = rounds dealt without
 *                         an atomic (0: the first, 1: all but the last, 2: half, 3: the first, and the chunks shrink towards the end of the batch).  Default 4 (first
 *                         round static, then four tiles per atomic) */
#define VPT_LAB_LANES 1u
#define VPT_LAB_LANE_GRID 2u
#define VPT_LAB_TAIL_GRID 3u
#define VPT_LAB_WHOLE_FRAMES 4u
#define VPT_LAB_WHOLE_SCHED 5u
int vpt_lab_set(vpt_ctx* ctx, uint32_t key, uint32_t value);
int vpt_lab_trace(vpt_ctx* ctx, uint32_t variant, uint32_t any_hit, const uint32_t* order_host, uint32_t param, uint32_t reps,
                  vpt_hit* hits_host, float* best_ms, uint64_t* visits);

#ifdef __cplusplus
}
#endif
#endif /* VPT_LAB_H */
