/*
 * rmd_api.h — C ABI of librmd.so, the MI355X (gfx950) filter / SVGF hot path.
 *
 * This is the drop-in boundary.  The reference (VictorHerbert/RaymarchDenoiserCuda) has no FFI
 * layer: its "API" is two __global__ symbols launched with <<<>>> straight from the test TU
 * (reference include/filter.cuh:25-26, src/test.cu:73-75,85-87).  Host code here stays plain
 * C/C++ and reaches the HIP kernels only through the extern "C" functions below; every function
 * names the reference interface it replaces.
 *
 * Conventions
 *   - All functions return 0 on success, a positive hipError_t on a HIP failure, or a negative
 *     RMD_E_* code on an argument error.  rmd_last_error_string() describes the last failure of
 *     the calling thread.  (The reference checks nothing: include/vector.h:119-169.)
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  Launchers are
 *     asynchronous on that stream.  The library never owns caller planes.
 *   - float4 planes are passed as `float*` (4 floats per pixel, 16-byte aligned, row-major,
 *     index y*W+x as in reference include/extended_math.h:66-68); float2 planes as `float*`
 *     (2 floats per pixel, 8-byte aligned).
 */
#ifndef RMD_API_H
#define RMD_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (negative = argument errors; positive = hipError_t) ------------------- */
#define RMD_OK              0
#define RMD_E_NULL         (-1)  /* a required pointer is NULL                               */
#define RMD_E_SHAPE        (-2)  /* non-positive width/height, or pixel count overflows int  */
#define RMD_E_PARAM        (-3)  /* radius<0, depth<1, unknown filter type, bad iteration ...*/
#define RMD_E_BUFFER       (-4)  /* depth>1 with NULL buffer[], or in/out planes alias       */
#define RMD_E_ROWS         (-5)  /* row range outside the frame or not covered by the buffer */
#define RMD_E_UNSUPPORTED  (-6)  /* filter type declared by the reference but not built yet  */
#define RMD_E_ALIGN        (-7)  /* plane pointer not aligned for its vector type            */
#define RMD_E_COMM         (-8)  /* RCCL is missing or a collective call failed              */

/* ---- PODs mirroring the reference host structs ------------------------------------------ */
typedef struct rmd_int2   { int x, y; } rmd_int2;
typedef struct rmd_uchar4 { unsigned char x, y, z, w; } rmd_uchar4;

/* reference include/gbuffer.h:6-14 — 56 bytes, align 8: shape@0 render@8 denoised@16 normal@24
 * albedo@32 buffer@40.  Passed BY VALUE exactly as the reference passes it in the kernarg. */
typedef struct rmd_gbuffer {
    rmd_int2    shape;
    rmd_uchar4* render;
    rmd_uchar4* denoised;
    rmd_uchar4* normal;
    rmd_uchar4* albedo;
    rmd_uchar4* buffer[2];
} rmd_gbuffer;

/* reference include/filter.cuh:11-23 — 36 bytes, align 4: type@0 depth@4 level@8 radius@12
 * sigmaSpace@16 sigmaColor@20 sigmaAlbedo@24 sigmaNormal@28 cacheInput@32 cacheBuffer@33.
 * (The reference struct has default member initialisers, so it is not valid C; this is its C
 * mirror and include/filter.h static_asserts that the two layouts agree.) */
enum { RMD_FILTER_AVERAGE = 0, RMD_FILTER_GAUSSIAN = 1, RMD_FILTER_CROSS = 2, RMD_FILTER_WAVELET = 3 };
typedef struct rmd_filter_params {
    int   type;
    int   depth;
    int   level;
    int   radius;
    float sigmaSpace;
    float sigmaColor;
    float sigmaAlbedo;
    float sigmaNormal;
    unsigned char cacheInput;   /* reference default: true */
    unsigned char cacheBuffer;  /* reference default: true */
} rmd_filter_params;

/* ---- uchar4 filter launchers (the path the reference actually implements) --------------- */

/* Replaces `filterKernelBaseline<<<grid,block,smem>>>(frame, params)` — reference
 * src/filter.cu:13-58 as launched at src/test.cu:73-75.  Box mean over the in-bounds taps of a
 * (2r+1)^2 window, fp32 accumulate, one divide, truncating cast; the output is GRAY from the R
 * mean (src/filter.cu:51-53).  One launch per level (the reference's single-launch level loop
 * races across blocks for depth>1).  .w of the output is written as 0. */
int rmd_filter_baseline(rmd_gbuffer frame, rmd_filter_params params, void* stream);

/* Replaces `filterKernelTiled<<<...>>>(frame, params)` — reference src/filter.cu:87-158 as
 * launched at src/test.cu:85-87.  Same box mean per RGB channel, .w = 0.  cacheInput=true stages
 * tile+halo in LDS (the job of reference cacheTile, src/filter.cu:60-85) with a consistent
 * stride, so both settings give the cacheInput=false result (the reference's cacheInput=true
 * output is corrupted by a stride mismatch, SURVEY §0.2, and is not a parity target).
 * params.type selects AVERAGE (reference behaviour) or the modes the reference declares but never
 * implements (include/filter.cuh:12-19; every kernel there uses w = 1):
 *   GAUSSIAN  w = exp(-(dx^2+dy^2)/(2 sigmaSpace^2)) over the (2r+1)^2 window; radius <= 127 (up to 12 on the separable LDS-tile
 *             kernel, beyond on a one-thread-per-pixel kernel that states the same operations: same bits, (2r+1)^2 gathers)
 *   CROSS     GAUSSIAN x exp(-|dc|^2/(2 sigmaColor^2)) x exp(-|da|^2/(2 sigmaAlbedo^2)) x
 *             exp(-|dn|^2/(2 sigmaNormal^2)) with c = the level's input plane, a / n = frame.albedo /
 *             frame.normal (0..255 units; a term with sigma <= 0 or a NULL plane is dropped)
 *   WAVELET   5x5 B3-spline taps {3/8,1/4,1/16} (src/filter.cu:10) at spacing 2^(level + l) for level
 *             index l, times the CROSS edge terms (edge-avoiding a-trous on the 8-bit planes)
 * Parity for these three is unpinned by the reference (csrc/weighted_filter.hip). */
int rmd_filter_tiled(rmd_gbuffer frame, rmd_filter_params params, void* stream);

/* ---- SVGF (north_star hot path; the reference only names it: README.md:3-10) ------------ */

/* Parameters of the three SVGF passes.  Semantics: SURVEY.md Appendix A (normative for this
 * build).  rmd_svgf_default_params() fills the defaults quoted there.
 *
 * Input contract.  nd.xyz of every pixel is a UNIT normal or exactly (0,0,0) ("no surface"; the
 * Cornell normal plane is 65 % zeros).  The kernels evaluate max(0, n.n')^sigma_n with
 * the cosine clamped to [0,1] and take "1 - n.n" as the test for a zero tap normal, so normals of
 * any other length get other weights in the a-trous passes than in the variance pass and the
 * oracle.  rmd_convert_u8_to_f32(renormalize_xyz = 1) and rmd_synth_gbuffer produce conforming
 * planes; CudaGBuffer::openImages uses the former.
 *
 * Conditioning (the one region where parity with the oracle is NOT claimed).  The luminance edge
 * weight is exp(-|dl| / (sigma_l * sqrt(var) + 1e-8)).  Where the variance channel is exactly 0 --
 * alpha_moments = 1 (moments never accumulate: m2 - m1^2 = 0), or var_radius = 0 on frames without
 * history -- the denominator is 1e-8 and the weight is a step function of the LAST BIT of dl: two
 * taps pass only if their luminances are equal to within ~1e-8, and whether they are is decided
 * by the rounding of the previous iteration (division vs reciprocal, order of summation).  Any two
 * correct implementations then differ on isolated pixels; this one stays within 5e-2 (1 + |ref|) of
 * the oracle there, with > 99.9 % of the values within the usual 5e-4 (1 + |ref|)
 * (tests/test_svgf_gpu.py::test_zero_variance_settings_are_fenced).  The settings are accepted:
 * the result is a valid edge-stopped filter, just not a reproducible one to the last digits. */
typedef struct rmd_svgf_params {
    /* T: temporal reprojection + accumulation */
    float alpha_color;      /* 0.05  minimum blend weight of the new colour sample             */
    float alpha_moments;    /* 0.2   minimum blend weight of the new luminance moments         */
    int   h_max;            /* 32    history length clamp, 1 .. 255 (the history length plane is uint8) */
    float k_z;              /* 10    depth-consistency slope factor                            */
    float k_n;              /* 0.9   normal-consistency cosine threshold                       */
    int   max_motion_rows;  /* 64    history taps with |tap.y - y| > this are invalid; makes the
                                     result independent of the row-strip decomposition         */
    /* V: spatial variance fallback */
    int   var_h_threshold;  /* 4     pixels with history < this use the 7x7 spatial estimate   */
    int   var_radius;       /* 3     */
    /* A: edge-stopping a-trous */
    float sigma_n;          /* 128 */
    float sigma_z;          /* 1   */
    float sigma_l;          /* 4   */
    int   iterations;       /* 5   step 2^i for i in [0, iterations)                           */
    int   hist_iteration;   /* 0   output of this iteration becomes next frame's hist_color    */
    int   atrous_variant;   /* 0 auto (= 3 for iterations 0..4, 1 beyond) | 1 direct (taps from global memory, any
                                   step; the cross-check) | 3 LDS row streaming, 128-column strips, two row pairs per
                                   workgroup.  0, 1 and 3 give identical bits.  Experiments build only
                                   (rmd_has_experiments(); RMD_E_UNSUPPORTED otherwise; measured slower, DESIGN.md
                                   §4.4-4.7): 2 / 6 row streaming with one / four row pairs per workgroup (same bits);
                                   pixel-pair formulation 4 LDS row streaming, 5 direct, 7 loader waves feeding
                                   compute waves through counters, 8 a 2x2 pixel block per lane (4, 5, 7, 8 agree
                                   with each other bit for bit and differ from 0/1/3 by rounding)              */
    int   tv_workgroups;    /* 0   T and V as one workgroup per 64x4 tile | N > 0 (experiments build only): N
                                   persistent workgroups that walk the tiles; same results, measured slower     */
    int   atrous_cus;       /* 0   CUs the a-trous launches may count on when they size their bands (0 = all CUs of
                                   the device; fewer when the caller knows other work holds part of the device)  */
    /* row strips across GPUs (SURVEY §8e "Halo sizes") */
    int   exchange_iteration; /* -1  every pass runs on the redundant rows the later passes tap, no exchange inside a frame |
                                   X in [0, iterations-2]: a strip computes iteration X on its OWN rows only and receives the
                                   2*(2^(X+1) + ... + 2^(iterations-1)) rows beyond them from rank +-1 (ONE neighbour exchange
                                   per frame); T, V and the iterations in front of X then run on that many fewer rows.
                                   Whole-frame calls are unaffected.  Strips drive the frame in parts
                                   (rmd_svgf_frame_atrous_part): iteration X's boundary rows first, its interior rows
                                   while the halo travels                                                       */
} rmd_svgf_params;

void rmd_svgf_default_params(rmd_svgf_params* p);

/* One frame of SVGF work on one device.  The planes may hold a ROW STRIP of a taller frame:
 * every plane stores global rows [buf_row0, buf_row0+buf_rows) of a width x height frame, local
 * row r <-> global row buf_row0+r.  Taps outside the GLOBAL frame are skipped and renormalised
 * (reference border rule, src/filter.cu:38-39,49); taps inside the global frame must be inside
 * the buffer (checked on the host, RMD_E_ROWS). */
typedef struct rmd_svgf_frame_desc {
    int width, height;        /* global frame */
    int buf_row0, buf_rows;   /* rows held by every plane below */
    /* inputs of this frame */
    const float* color;        /* float4 rgb = noisy illumination, a = unused on input          */
    const float* nd;           /* float4 (nx,ny,nz, linear z)                                   */
    const float* motion;       /* float2 pixels, current -> previous                            */
    /* history (previous frame) */
    const float* hist_color;   /* float4 rgb + variance                                         */
    const float* hist_moments; /* float2 (m1, m2): first and second luminance moment, 8-byte aligned       */
    const unsigned char* hist_len; /* uint8 history length h (1 .. h_max <= 255), one byte per pixel.  The moments used to
                                  be a float4 (m1, m2, h, 0): 16 B/px read per reprojection tap and 16 written, of which 9 carry
                                  anything -- the temporal pass is HBM-bound and moved 120 B/px, now 106              */
    const float* prev_nd;      /* float4                                                        */
    /* intermediates / outputs */
    float* t_color;            /* T out: float4 (c', variance).  rmd_svgf_frame / rmd_svgf_frame_tv with v_tile_flags set
                                  treat it as scratch: only the 64x4 tiles T flags for V are written (v_color holds
                                  T's output for every pixel); rmd_svgf_temporal always writes all of it            */
    float* t_moments;          /* T out: float2 (m1', m2') -> next frame's hist_moments         */
    unsigned char* t_len;      /* T out: uint8 h -> next frame's hist_len                       */
    int*   t_debug;            /* optional int4 (q0.x, q0.y, tap mask, h): the bit-exact outputs */
    float* v_color;            /* V out: float4                                                 */
    float* hist_color_out;     /* A out of iteration `hist_iteration` -> next frame's hist_color */
    float* ping[2];            /* A ping-pong planes                                            */
    float* out_color;          /* A out of the last iteration: denoised illumination + variance */
    float* stats;              /* optional 4 floats, ACCUMULATED by V with wavefront reductions:
                                  [0] sum of variance, [1] pixels on the spatial path,
                                  [2] sum of history length, [3] pixels processed              */
    unsigned char* v_tile_flags; /* optional scratch used by rmd_svgf_frame: one byte per 64x4-pixel tile of
                                  the GLOBAL frame, RMD_TILE_FLAGS_BYTES(width, height) bytes, 4-byte aligned.  T marks
                                  the tiles that contain short-history pixels, so V skips every other
                                  tile without reading a byte of it.  NULL = V visits all pixels.     */
} rmd_svgf_frame_desc;
#define RMD_TILE_FLAGS_BYTES(width, height) (((size_t)(((width) + 63) / 64) * (size_t)(((height) + 3) / 4) + 3) / 4 * 4)

/* Pass launchers.  [row0,row1) are GLOBAL output rows. */
int rmd_svgf_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream);
int rmd_svgf_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream);
/* One a-trous iteration (step 2^iteration) from plane `in` to plane `out` (both float4, same
 * buffer geometry as f; f->nd supplies normals/depth). */
int rmd_svgf_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                    const float* in, float* out, int row0, int row1, void* stream);
/* The same iteration on TWO row ranges, [row0,row1) and [row0b,row1b) behind it, in ONE launch (the two boundary bands of a
 * strip's exchanged iteration: as two launches of one step each they cost twice the fixed part of a launch).  Same bits
 * as two calls.  row0b >= row1b: no second range. */
int rmd_svgf_atrous2(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                     const float* in, float* out, int row0, int row1, int row0b, int row1b, void* stream);
/* Variant 7 synchronises its waves with counters in LDS; every wait is bounded, and a wait that runs out
 * is counted here instead of hanging the GPU (synchronises the device; 0 after any number of correct launches). */
int rmd_debug_atrous_protocol_errors(unsigned int* count);
/* The work decomposition the default a-trous kernel (128-column strips) would use for rows [row0,row1) of a
 * width x height frame on `cus` CUs; host arithmetic only, no device needed.  out[8] = workgroups, strips, first band
 * row, band height, band height of the strips that get one band more, number of those strips (the outermost: the first
 * n/2 and the last n - n/2), workgroups of those strips (they come first in the order), workgroup slots per XCD.
 * Workgroup L (< out[0]): lattice L % step; strip group and band as in atrous_stream_kernel.  For tests.          */
int rmd_debug_atrous_plan(int width, int height, int row0, int row1, int iteration, int cus, int* out);
/* T + V + `iterations` x A for final output rows [row0,row1).  Earlier passes are computed on
 * the rows later passes tap (redundant rows instead of per-pass halo exchanges);
 * those rows are clamped to the global frame and must lie inside the buffer. */
int rmd_svgf_frame(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream);
/* The two halves of rmd_svgf_frame, for software pipelining across frames: T+V of frame k+1 only
 * needs frame k's history, which is complete after frame k's `hist_iteration`, so it can run on a
 * second stream underneath frame k's remaining a-trous iterations (T is HBM-bound, A is ALU-bound).
 * rmd_svgf_frame_atrous records `history_ready_event` (a hipEvent_t, may be NULL) on `stream` right
 * after the hist_iteration launch. */
int rmd_svgf_frame_tv(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream);
int rmd_svgf_frame_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                          void* history_ready_event);
/* Frame pipelining INSIDE the a-trous waves: the iterations of frame `f` behind hist_iteration carry the temporal pass of
 * frame `next` as a side job (one 64x4 tile per workgroup every few steps; T is HBM-bound with a half idle VALU, the iterations
 * are VALU-bound with HBM at 40 %, and as a side job T needs no registers of its own beside three a-trous workgroups per CU);
 * the tiles they leave over and V(next) follow as launches of their own.  On return (in stream order) `next` is where
 * rmd_svgf_frame_tv(next, ...) would have left it -- same kernel code, same bits -- and the caller continues with
 * rmd_svgf_frame_atrous[_next](next, ...).  `next` names the planes of the following frame: hist_color = f->hist_color_out,
 * hist_moments = f->t_moments, prev_nd = f->nd; the planes T / V(next) write (t_color, t_moments, v_color, t_debug) must be none
 * of f's ping / out_color / hist_color_out / nd.  Whole frames or strips without a mid-frame exchange, no statistics.
 * MEASURED AND LOST (4K: 0.999 ms per frame against 0.877-0.906 serial): part of the experiments build
 * only (rmd_has_experiments()); the product library returns RMD_E_UNSUPPORTED. */
int rmd_svgf_frame_atrous_next(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                               void* history_ready_event, const rmd_svgf_frame_desc* next);
/* The a-trous iterations of a strip whose params name an exchange_iteration X, in the three parts the exchange cuts them into:
 *   RMD_ATROUS_HEAD      iterations 0..X-1, then iteration X on the BOUNDARY rows of the strip: the rows rank +-1 is waiting
 *                        for (rmd_svgf_frame_mid_exchange()[1] rows at each end that is not a frame edge).  Then the caller
 *                        starts the neighbour exchange of the plane rmd_svgf_frame_iteration_plane(f, p, X) on another
 *                        stream, ordered behind an event recorded here;
 *   RMD_ATROUS_INTERIOR  iteration X on the remaining rows of the strip: runs while the halo travels;
 *   RMD_ATROUS_TAIL      after the exchange has completed: iterations X+1 ... (whole launches, halo rows included)
 *   RMD_ATROUS_ALL       everything in order (= rmd_svgf_frame_atrous).  Only where no halo is needed: whole frames, or
 *                        exchange_iteration = -1; a strip with exchange_iteration >= 0 is refused (RMD_E_PARAM): the
 *                        iteration behind the exchanged one would read halo rows nobody delivered.
 * Same kernels on other row ranges: bit-identical to the unsharded frame (tests/test_svgf_gpu.py, test_sharding_*). */
enum { RMD_ATROUS_ALL = 0, RMD_ATROUS_HEAD = 1, RMD_ATROUS_INTERIOR = 2, RMD_ATROUS_TAIL = 3 };
int rmd_svgf_frame_atrous_part(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                               void* history_ready_event, int part);
/* reach[i] = rows above/below [row0,row1) on which a strip computes a-trous iteration i (0 beyond `iterations`). */
int rmd_svgf_frame_iteration_reach(const rmd_svgf_params* p, int reach[8]);
/* mid[0] = exchange_iteration (or -1), mid[1] = rows per side that travel in the mid-frame exchange. */
int rmd_svgf_frame_mid_exchange(const rmd_svgf_params* p, int mid[2]);
/* The plane iteration `iteration` writes under rmd_svgf_frame's routing (ping[] / hist_color_out / out_color). */
int rmd_svgf_frame_iteration_plane(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration, float** plane);
/* Rows needed / produced above and below [row0,row1) by rmd_svgf_frame (for sizing buffers and
 * halo exchanges): reach[0] = current-frame input planes read, reach[1] = history planes read,
 * reach[2] = rows on which hist_color_out is (redundantly) produced, reach[3] = same for
 * t_moments.  A strip already holds reach[2]/reach[3] halo rows of its own next-frame history
 * bit-identical to its neighbour's, so only rows (reach[2], reach[1]] resp. (reach[3], reach[1]]
 * have to be exchanged. */
int rmd_svgf_frame_reach(const rmd_svgf_params* p, int reach[4]);

/* Opaque per-device context owning the cross-frame state (history planes are the only state
 * that outlives a frame; SURVEY §5 "Checkpoint / resume"). */
typedef struct rmd_svgf_context rmd_svgf_context;
int  rmd_svgf_context_create(int width, int height, int buf_row0, int buf_rows, rmd_svgf_context** out);
void rmd_svgf_context_destroy(rmd_svgf_context* ctx);
int  rmd_svgf_context_reset_history(rmd_svgf_context* ctx, void* stream);
/* Denoise one frame: borrows color/nd/motion/prev_nd (prev_nd = the nd plane passed for the
 * previous frame; the caller double-buffers its G-buffer), writes rows [row0,row1) of `out`
 * (float4, same buffer geometry) and rotates the history planes. */
int  rmd_svgf_context_denoise(rmd_svgf_context* ctx, const rmd_svgf_params* p,
                              const float* color, const float* nd, const float* motion,
                              const float* prev_nd, float* out, int row0, int row1, void* stream);
/* The same frame in the parts of rmd_svgf_frame_atrous_part: RMD_ATROUS_HEAD runs T + V + iterations 0..X-1 + the boundary
 * rows of X, RMD_ATROUS_INTERIOR / RMD_ATROUS_TAIL the rest; the history planes rotate with the TAIL part.  Between HEAD and
 * TAIL the caller exchanges the rows of rmd_svgf_context_mid_plane with rank +-1 (rmd_mid_exchange).  RMD_ATROUS_ALL is
 * rmd_svgf_context_denoise (T + V + every iteration + the rotation). */
int  rmd_svgf_context_denoise_part(rmd_svgf_context* ctx, const rmd_svgf_params* p,
                                   const float* color, const float* nd, const float* motion,
                                   const float* prev_nd, float* out, int row0, int row1, void* stream, int part);
int  rmd_svgf_context_mid_plane(rmd_svgf_context* ctx, const rmd_svgf_params* p, float** plane);
/* SVGF straight on the reference's frame descriptor (include/gbuffer.h:6-14; CudaGBuffer::openImages, :20-33, is the
 * declared hook that fills it): ONE call per frame, uchar4 `render` / `albedo` / `normal` in, uchar4 `denoised` out, six kernel
 * launches.  It computes exactly what this chain of eight calls computes, to the byte (tests/test_gbuffer_frame_gpu.py):
 *     rmd_convert_u8_to_f32(render, color, n, 0, 0)        rmd_convert_u8_to_f32(albedo, alb, n, 0, 0)
 *     rmd_convert_u8_to_f32(normal, nd, n, 1, -1)          rmd_demodulate(color, alb, color, n, albedo_eps)
 *     rmd_svgf_context_denoise(ctx, p, color, nd, motion, previous nd, out, 0, height, stream)
 *     rmd_convert_f32_to_u8(out, alb, denoised, n)
 * but the 8-bit ends run INSIDE the frame's first and last launch: the T+V kernel reads the three uchar4 planes (12 B/px instead
 * of 32 B/px of float planes), converts, renormalises and demodulates in registers and writes the float (normal, depth) plane
 * once (the a-trous passes and the next frame's reprojection read it; the context owns two of them); the last a-trous launch
 * multiplies by the albedo, quantises and stores bytes (4 B/px written instead of 16).  The chain's five conversion launches
 * and their 144 B/px do not exist.
 *   normal.xyz  world normal in [0,1] as the reference's fixtures hold it, renormalised to unit length; (0,0,0) = no surface
 *   normal.w    linear depth in units of 1/255.  The reference's GBuffer has no depth plane and its Image(path, 4) loader
 *               yields an opaque alpha (255): depth 1 everywhere, which is also all the saturated depth.png of the fixture says
 *   motion      float2 per pixel (current -> previous, pixels) or NULL = static camera
 *   albedo_eps  > 0: floor of the demodulation's denominator (black albedo)
 * `ctx` must hold whole frames of frame.shape; rmd_svgf_params.exchange_iteration must be -1, var_radius 3 (the front end lives in
 * the fused T+V launch; RMD_E_UNSUPPORTED otherwise: take the float planes).  frame.buffer[] is not used.  Frames of this
 * call and of rmd_svgf_context_denoise may alternate on one context; the first frame after a switch starts a new history. */
int  rmd_svgf_gbuffer_frame(rmd_gbuffer frame, rmd_svgf_context* ctx, const rmd_svgf_params* p, const float* motion,
                            float albedo_eps, void* stream);
/* Optional int4 plane (q0.x, q0.y, tap mask, h per pixel, whole buffer) that T of every following frame of this context
 * fills: the bit-exact outputs, for tests.  NULL switches it off. */
int  rmd_svgf_context_set_debug_plane(rmd_svgf_context* ctx, int* t_debug);
/* The history planes the NEXT rmd_svgf_context_denoise call will read (for halo exchange). */
int  rmd_svgf_context_history(rmd_svgf_context* ctx, float** hist_color, float** hist_moments, unsigned char** hist_len);
/* Fill a descriptor with the context's planes for the next frame (advanced use / tests). */
int  rmd_svgf_context_describe(rmd_svgf_context* ctx, rmd_svgf_frame_desc* f);

/* ---- row strips across the GPUs of a node (the reference is single-device) ----- */
/* A frame of `height` rows is cut into `world` contiguous full-width row strips, one per GPU.  A rank
 * runs rmd_svgf_frame on its strip plus the rows later passes tap (redundant rows, no exchange inside a
 * frame); only the temporal feedback crosses ranks: the history rows beyond what a rank produces
 * itself, point to point with rank +- 1.  The planning functions are pure arithmetic (no device). */
typedef struct rmd_strip_plan {
    int height, world, rank;
    int row0, row1;            /* owned output rows [row0,row1)                                    */
    int buf_row0, buf_rows;    /* rows every plane of this rank holds (strip + reach, clamped)      */
    int reach_in, reach_hist;  /* rmd_svgf_frame_reach()[0], [1]                                     */
    int have_color, have_moments; /* ...[2], [3]: history rows this rank produces beyond its strip   */
    int mid_iteration, mid_rows;  /* rmd_svgf_frame_mid_exchange(): the a-trous iteration whose output crosses ranks inside
                                     a frame (-1 = none) and the rows per side that travel                  */
} rmd_strip_plan;
enum { RMD_HALO_RECV = 0, RMD_HALO_SEND = 1, RMD_HALO_MAX_STEPS = 12 };
/* planes a halo step may name, and their bytes per pixel: 16 (float4), 8 (float2), 16 (float4), 1 (uint8) */
enum { RMD_PLANE_HIST_COLOR = 0, RMD_PLANE_HIST_MOMENTS = 1, RMD_PLANE_MID = 2, RMD_PLANE_HIST_LEN = 3, RMD_PLANE_COUNT = 4 };
typedef struct rmd_halo_step {
    int kind;                  /* RMD_HALO_RECV | RMD_HALO_SEND                                      */
    int plane;                 /* RMD_PLANE_HIST_COLOR | RMD_PLANE_HIST_MOMENTS | RMD_PLANE_HIST_LEN | RMD_PLANE_MID   */
    int row_lo, row_hi;        /* GLOBAL rows [row_lo,row_hi)                                        */
    int peer;                  /* rank - 1 or rank + 1                                               */
} rmd_halo_step;
int    rmd_strip_rows(int height, int world, int rank, int* row0, int* row1);
/* RMD_E_ROWS if a strip would be shorter than the history reach (use fewer ranks). */
int    rmd_strip_plan_make(int height, int world, int rank, const rmd_svgf_params* p, rmd_strip_plan* out);
/* The exchange of one rank and frame.  Receives and the neighbour's matching sends appear in the same
 * order, so posting them as one group cannot deadlock.  steps may be NULL to count. */
int    rmd_halo_plan(const rmd_strip_plan* plan, rmd_halo_step* steps, int max_steps, int* n_steps);
size_t rmd_halo_bytes(const rmd_strip_plan* plan, int width);   /* history bytes this rank receives per frame */
/* The exchange INSIDE a frame (rmd_svgf_params.exchange_iteration = X >= 0): plan->mid_rows rows of iteration X's output on
 * either side of the strip, received from the neighbour that computed them as its own rows (plane RMD_PLANE_MID). */
int    rmd_mid_halo_plan(const rmd_strip_plan* plan, rmd_halo_step* steps, int max_steps, int* n_steps);

/* RCCL communicator (librccl.so is opened on first use; rmd_comm_available() == 0 without it).
 * Multi-process: rank 0 calls rmd_comm_unique_id and hands the 128 bytes to the other ranks (the host
 * program's job: a file, a socket, MPI); every rank calls rmd_comm_create on its own device.
 * Single process, several GPUs: rmd_comm_create_all (ncclCommInitAll). */
typedef struct rmd_comm rmd_comm;
int rmd_comm_available(void);
int rmd_comm_unique_id(void* id128);
int rmd_comm_create(const void* id128, int world, int rank, rmd_comm** out);
int rmd_comm_create_all(int ndev, const int* devices /* NULL = 0..ndev-1 */, rmd_comm** out);
int rmd_comm_destroy(rmd_comm* c);
/* The per-frame history halo exchange of this rank: ncclGroupStart; ncclSend / ncclRecv(rank +- 1) of
 * the rmd_halo_plan rows (the rows named for RMD_PLANE_HIST_MOMENTS travel for RMD_PLANE_HIST_LEN as well: the plan lists
 * both); ncclGroupEnd -- asynchronous on `stream`.  Call it between frame k's
 * hist_iteration and frame k+1's temporal pass, on the planes rmd_svgf_context_history returns.
 * world == 1 (or an empty plan) is a no-op and needs no communicator. */
int rmd_halo_exchange(rmd_comm* c, const rmd_strip_plan* plan, int width, float* hist_color, float* hist_moments,
                      unsigned char* hist_len, void* stream);
/* Single-process form: plans[k], planes and stream of rank k for k < world, all ranks in one group. */
int rmd_halo_exchange_all(rmd_comm* c, const rmd_strip_plan* plans, int width, float* const* hist_color,
                          float* const* hist_moments, unsigned char* const* hist_len, void* const* streams);
/* The mid-frame exchange (rmd_mid_halo_plan) of iteration X's output plane -- rmd_svgf_context_mid_plane /
 * rmd_svgf_frame_iteration_plane -- between RMD_ATROUS_HEAD and RMD_ATROUS_TAIL, asynchronous on `stream` (a second
 * stream, so that RMD_ATROUS_INTERIOR runs meanwhile). */
int rmd_mid_exchange(rmd_comm* c, const rmd_strip_plan* plan, int width, float* mid_plane, void* stream);
int rmd_mid_exchange_all(rmd_comm* c, const rmd_strip_plan* plans, int width, float* const* mid_planes, void* const* streams);
/* Explicit steps on communicator comm_index of c (tests: a loop-back exchange on one GPU). */
int rmd_halo_exchange_steps(rmd_comm* c, int comm_index, const rmd_halo_step* steps, int n_steps, int buf_row0, int buf_rows,
                            int width, float* hist_color, float* hist_moments, unsigned char* hist_len, void* stream);
/* The same with all planes a step may name: planes[RMD_PLANE_HIST_COLOR], [RMD_PLANE_HIST_MOMENTS], [RMD_PLANE_MID],
 * [RMD_PLANE_HIST_LEN] (entries no step names may be NULL). */
int rmd_exchange_steps(rmd_comm* c, int comm_index, const rmd_halo_step* steps, int n_steps, int buf_row0, int buf_rows,
                       int width, void* const planes[RMD_PLANE_COUNT], void* stream);

/* ---- 8-bit <-> float plane conversion --------------------------------- */
/* uchar4 -> float4, c/255; optional per-pixel renormalisation of xyz (for normals). */
int rmd_convert_u8_to_f32(const rmd_uchar4* in, float* out, size_t pixels, int renormalize_xyz, float w_value, void* stream);
/* float4 illumination (x optional float4 albedo) -> clamp -> uchar4. */
int rmd_convert_f32_to_u8(const float* in, const float* albedo, rmd_uchar4* out, size_t pixels, void* stream);
/* Demodulation by albedo BEFORE filtering (the render / albedo planes of reference include/gbuffer.h:10-12
 * imply the split): out.rgb = radiance.rgb / max(albedo.rgb, eps), out.w = radiance.w.  SVGF then filters
 * illumination and rmd_convert_f32_to_u8(..., albedo, ...) multiplies the albedo back.  in == out is allowed. */
int rmd_demodulate(const float* radiance, const float* albedo, float* out, size_t pixels, float eps, void* stream);

/* ---- synthetic G-buffer generator (SURVEY §8d "Synthetic inputs") ------------------------ */
typedef struct rmd_synth_desc {
    int      width, height;     /* global frame */
    int      buf_row0, buf_rows;
    uint32_t seed;              /* 1234 */
    int      frame;             /* 0..59 */
    float    pan_x, pan_y;      /* 1.25, -0.5 px/frame */
} rmd_synth_desc;
/* Writes color (float4), nd (float4), motion (float2), albedo (float4, optional) for the
 * buffer rows; counter-based RNG, so every rank generates identical pixels. */
int rmd_synth_gbuffer(const rmd_synth_desc* d, float* color, float* nd, float* motion, float* albedo, void* stream);

/* ---- device memory / runtime (replaces reference CudaVector internals, vector.h:119-169) - */
int  rmd_malloc(void** ptr, size_t bytes);
int  rmd_free(void* ptr);
int  rmd_memset(void* ptr, int value, size_t bytes, void* stream);
int  rmd_memcpy_h2d(void* dst, const void* src, size_t bytes);
int  rmd_memcpy_d2h(void* dst, const void* src, size_t bytes);
int  rmd_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int  rmd_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream);
int  rmd_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream);
int  rmd_host_alloc_pinned(void** ptr, size_t bytes);
int  rmd_host_free_pinned(void* ptr);
int  rmd_stream_create(void** stream);
int  rmd_stream_destroy(void* stream);
int  rmd_stream_sync(void* stream);
/* events order work across streams (frame pipelining): record on one stream, wait on another */
int  rmd_event_create(void** event);
int  rmd_event_destroy(void* event);
int  rmd_event_record(void* event, void* stream);
int  rmd_event_synchronize(void* event);   /* host waits for the event (e.g. an upload queued by openImages) */
int  rmd_stream_wait_event(void* stream, void* event);
/* hipGraph capture of whatever the caller queues on `stream` between begin and end (no counterpart in the reference: it
   launches on the default stream, src/test.cu:76,88).  Every rmd_* launch entry point may be captured once it has run
   eagerly on the device (the first call sets kernel attributes).  For a host whose own launch path is slow; the frame loop
   of this package gains nothing from it (its six launches are queued ahead of the GPU anyway: 1080p 0.276 / 0.287 ms per
   frame replayed against 0.287 / 0.271 eager in two runs, 4K 0.877 against 0.874, tools/graph_probe.py).  The graph holds
   the POINTERS of the captured calls: capture an even number of frames of an rmd_svgf_context / SvgfDenoiser (its history
   planes ping-pong) and keep the input planes in place; it also bakes in whether a frame had a history and the
   parameters: no rmd_svgf_context_reset_history and no other rmd_svgf_params between capture and replay.
   `stream` must be a created stream, not NULL (the legacy default stream cannot be captured). */
int  rmd_graph_capture_begin(void* stream);
int  rmd_graph_capture_end(void* stream, void** graph);     /* ends the capture, instantiates; on failure the capture is ended and *graph = NULL */
int  rmd_graph_launch(void* graph, void* stream);
int  rmd_graph_destroy(void* graph);                         /* NULL allowed */
int  rmd_device_sync(void);              /* reference cudaDeviceSynchronize(), src/test.cu:77,89 */
int  rmd_device_count(int* count);
int  rmd_set_device(int device);
/* reference printGPUProperties(), src/utils.cpp:5-15 */
int  rmd_print_device_properties(void);
const char* rmd_last_error_string(void);
const char* rmd_version(void);
/* 1 if the library was built with -DRMD_EXPERIMENTS (`make experiments`: the a-trous formulations that were measured
 * and lost -- atrous_variant 2, 4, 5, 6, 7, 8 --, the persistent T / V kernels behind tv_workgroups, and the
 * environment tuning knobs of the measurement tools), 0 for the product library, where those settings return
 * RMD_E_UNSUPPORTED. */
int  rmd_has_experiments(void);

/* HIP-event timing of a region on a stream (bench / harness). */
int  rmd_timer_create(void** timer);
int  rmd_timer_destroy(void* timer);
int  rmd_timer_start(void* timer, void* stream);
int  rmd_timer_stop(void* timer, void* stream);
int  rmd_timer_elapsed_ms(void* timer, float* ms); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* RMD_API_H */
