/*
 * jrender_hip.h — C ABI of the MI355X-native SoftRas hot path (libjrender_hip.so).
 *
 * Drop-in boundary for jrender's `soft_rasterize` operator.  The reference binds
 * its kernels through Jittor's JIT operator `jt.code(out_shapes, out_dtypes,
 * inputs, cuda_header, cuda_src)`:
 *   forward  op: jrender/renderer/dr/softras/cuda/soft_rasterize.py:3-521
 *                inputs {faces, textures} -> outputs {faces_info, aggrs_info,
 *                soft_colors, faces_id_buffer}
 *   backward op: jrender/renderer/dr/softras/cuda/soft_rasterize.py:966-1416
 *                inputs {faces, textures, soft_colors, faces_info, aggrs_info,
 *                grad_soft_colors, faces_id_buffer} -> {grad_faces, grad_textures}
 * called from SoftRasterizeFunction.execute / .grad
 * (jrender/renderer/dr/softras/soft_rasterize.py:34-103, :105-133).
 * The entry points below are what an FFI for those two ops binds instead
 * (ctypes stub: INTEGRATION.md).  Plain pointers and sizes only; every array
 * pointer is a DEVICE pointer on the context's GPU unless marked "host".
 *
 * Conventions
 *   - all functions return 0 on success, non-zero on failure; jr_last_error()
 *     returns a thread-local message for the last failure.  (The reference only
 *     printf's launch errors, SRK:487-489.)
 *   - work is enqueued on the context's HIP stream; results are complete after
 *     jr_synchronize() or any blocking copy.
 *   - tensor layouts and dtypes are exactly the reference's (row-major, fp32 /
 *     int32), EXCEPT that the backward takes faces_id_buffer in the forward's own
 *     [B,K,IS,IS] layout: the reference's extra transpose to [B,IS,IS,K]
 *     (soft_rasterize.py:108) is pure re-indexing and is skipped.
 *   - scalars are passed as float exactly like the literals the reference
 *     interpolates into its launches (SRK:485-516); `dist_eps` is the already
 *     transformed value log(1/dist_eps - 1) of soft_rasterize.py:25.
 */
#ifndef JRENDER_HIP_H
#define JRENDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jr_ctx jr_ctx; /* one per (process, GPU): device id, stream, scratch arena */

/* enum ids = the reference's maps, soft_rasterize.py:39-42 */
enum { JR_DIST_HARD = 0, JR_DIST_BARYCENTRIC = 1, JR_DIST_EUCLIDEAN = 2 };
enum { JR_RGB_HARD = 0, JR_RGB_SOFTMAX = 1, JR_RGB_NONE = 2 };
enum { JR_ALPHA_HARD = 0, JR_ALPHA_SUM = 1, JR_ALPHA_PROD = 2 };
enum { JR_TEX_SURFACE = 0, JR_TEX_VERTEX = 1 };

#define JR_MAX_FACES_PER_PIXEL 64 /* reference: kMaxPointsPerPixel, SRK:16 (unchecked there) */

/* ---- runtime / memory (replaces what the Jittor runtime provided) ---------- */
const char* jr_last_error(void);
const char* jr_version(void);
int jr_device_count(int* count);
int jr_ctx_create(int device, jr_ctx** out);
int jr_ctx_destroy(jr_ctx* ctx);
int jr_ctx_device(const jr_ctx* ctx);
void* jr_ctx_stream(const jr_ctx* ctx); /* hipStream_t */
int jr_malloc(jr_ctx* ctx, size_t bytes, void** dptr);
int jr_free(jr_ctx* ctx, void* dptr);   /* returns the block to the context's cache (stream-ordered reuse) */
int jr_ctx_trim(jr_ctx* ctx);           /* hipFree everything cached by jr_free */
int jr_memcpy_h2d(jr_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* blocking */
int jr_memcpy_d2h(jr_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* blocking */
int jr_memcpy_d2d(jr_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);  /* async   */
/* `height` rows of `width` bytes, rows `*_pitch` bytes apart (channel selection out of [B,4,H,W] images); async */
int jr_memcpy2d_d2d(jr_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t width,
                    size_t height);
int jr_memset(jr_ctx* ctx, void* dptr, int byte_value, size_t bytes);              /* async   */
int jr_synchronize(jr_ctx* ctx);
/* HIP events on the context's stream (benchmark timing) */
int jr_event_create(jr_ctx* ctx, void** event);
int jr_event_destroy(jr_ctx* ctx, void* event);
int jr_event_record(jr_ctx* ctx, void* event);
int jr_event_elapsed_ms(jr_ctx* ctx, void* start, void* stop, float* ms); /* syncs on stop */

/* ---- SoftRas forward: replaces forward_soft_rasterize (SRK:3-521) ---------------
 * in : face_vertices [B,NF,9] f32 (NDC x,y ; camera z per vertex), textures [B,NF,T,3] f32
 * out: faces_info [B,NF,27] f32, aggrs_info [B,2,IS,IS] f32, soft_colors [B,4,IS,IS] f32,
 *      faces_id_buffer [B,K,IS,IS] i32 (-1 = empty slot).  Outputs need no pre-initialisation.
 * background_rgb: host pointer to 3 floats, or NULL = the reference's behaviour (background
 *      colour ignored, i.e. 0; soft_rasterize.py:68-74 vs SRK:469).
 * K = max_faces_per_pixel_for_grad, 1..JR_MAX_FACES_PER_PIXEL.  IS <= 4096.  NF <= 2^28 - 1 faces per image,
 * B * NF <= 2^31 - 1 (the face records and the bin lists index with these widths); violations return non-zero.
 */
int jr_softras_forward(jr_ctx* ctx, const float* face_vertices, const float* textures,
                       float* faces_info, float* aggrs_info, float* soft_colors,
                       int32_t* faces_id_buffer, int B, int NF, int T, int IS, int K,
                       float near_, float far_, float eps, float sigma_val, int func_id_dist,
                       float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                       int texture_sample_type, int double_side, const float* background_rgb);

/* ---- SoftRas backward: replaces backward_soft_rasterize (SRK:966-1416) ----------
 * in : the forward's inputs and outputs + grad_soft_colors [B,4,IS,IS] f32
 * out: grad_faces [B,NF,9] f32, grad_textures [B,NF,T,3] f32 (zeroed here, like SRK:1374-1375)
 */
int jr_softras_backward(jr_ctx* ctx, const float* face_vertices, const float* textures,
                        const float* soft_colors, const float* faces_info,
                        const float* aggrs_info, const int32_t* faces_id_buffer,
                        const float* grad_soft_colors, float* grad_faces, float* grad_textures,
                        int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                        float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                        int func_id_rgb, int func_id_alpha, int texture_sample_type,
                        int double_side);

/* The backward needs the per-face records and the launch order that the forward's set-up pass built.
 * jr_softras_backward always REBUILDS them from face_vertices / textures as passed (0.06 ms on the
 * headline workload).  A caller that runs the backward of the context's LATEST forward can skip that:
 * jr_softras_forward_token() returns a non-zero generation token right after a successful forward, and
 * jr_softras_backward_ex() reuses the forward's records iff that token is still the context's current
 * generation (no other forward / backward set-up ran on the context in between) and the shapes match.
 * The token says nothing about the CONTENT of face_vertices / textures: passing it asserts that the
 * caller has not modified those buffers since the forward (the Python mirror clones them like the
 * reference, soft_rasterize.py:59-60).  Token 0 = always rebuild.  Pointer identity is never used. */
uint64_t jr_softras_forward_token(const jr_ctx* ctx);
int jr_softras_backward_ex(jr_ctx* ctx, const float* face_vertices, const float* textures,
                           const float* soft_colors, const float* faces_info,
                           const float* aggrs_info, const int32_t* faces_id_buffer,
                           const float* grad_soft_colors, float* grad_faces, float* grad_textures,
                           int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                           float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                           int func_id_rgb, int func_id_alpha, int texture_sample_type,
                           int double_side, uint64_t forward_token);

/* ---- adjacent steps the reference ran as Jittor tensor ops -----------------------
 * face_vertices gather  vertices[B,NV,3] x faces[NF,3] (shared) -> [B,NF,9]
 *   (jrender/structures/utils/faces_vertices.py:4-19) and its scatter-add backward
 *   (was Jittor autograd).
 * 2x2 mean pool for anti_aliasing (nn.pool(images, 2, "mean", stride=2),
 *   jrender/renderer/dr/softras/rasterizer.py:54-55) and its backward.
 */
int jr_face_vertices_forward(jr_ctx* ctx, const float* vertices, const int32_t* faces,
                             float* face_vertices, int B, int NV, int NF);
int jr_face_vertices_backward(jr_ctx* ctx, const float* grad_face_vertices, const int32_t* faces,
                              float* grad_vertices, int B, int NV, int NF);
/* views that share ONE vertex set (demo2-deform.py:45): sum of the per-view scatter-adds -> [NV,3] */
int jr_face_vertices_backward_shared(jr_ctx* ctx, const float* grad_face_vertices,
                                     const int32_t* faces, float* grad_vertices, int B, int NV,
                                     int NF);
/* ---- the callers either side of the op, device-resident (BASELINE configs[3]) ----------
 * Camera step of camera_mode 'look_at' / 'look' (jrender/renderer/transform/look_at.py:3-39, look.py:3-54) followed by
 * perspective (perspective.py:4-17, kind 1, param = tan(viewing_angle)), orthogonal (orthogonal.py:3-16, kind 2,
 * param = scale) or nothing (kind 0).  eye [B,3] and rot [B,9] (rows = camera x, y, z axes) are computed by the host
 * (O(B)); vertices are [VB,NV,3] with VB == B, or VB == 1 for ONE vertex set seen by all B views (what
 * demo2-deform.py:45 builds with repeat()).  out [B,NV,3].  The backward (was Jittor autograd) returns
 * d/d(world vertices) [VB,NV,3]; with VB == 1 it is the sum over the views, added in view order (no atomics). */
int jr_camera_forward(jr_ctx* ctx, const float* vertices, const float* eye, const float* rot, float* out,
                      int B, int VB, int NV, int kind, float param);
int jr_camera_backward(jr_ctx* ctx, const float* grad_out, const float* vertices, const float* eye,
                       const float* rot, float* grad_vertices, int B, int VB, int NV, int kind, float param);
/* jr_face_vertices_backward_shared and jr_camera_backward (VB == 1) in one pass: grad_face_vertices [B,NF,9] in NDC
 * space -> d/d(the ONE world-space vertex set) [NV,3], summed over faces and views (float atomics per face corner). */
int jr_face_camera_backward_shared(jr_ctx* ctx, const float* grad_face_vertices, const int32_t* faces,
                                   const float* vertices, const float* eye, const float* rot, float* grad_vertices,
                                   int B, int NV, int NF, int kind, float param);
/* neg_iou_loss (jrender/loss/iou_loss.py:1-9) per view: iou[b] = sum(p*t) / (sum(p + t - p*t) + 1e-6) over the n
 * elements of view b (the loss is 1 - mean(iou)); grad_predict (may be NULL) = d(loss)/d(predict) with the mean
 * taken over `divisor` views (the whole batch when the views are sharded over ranks). */
int jr_neg_iou_loss(jr_ctx* ctx, const float* predict, const float* target, float* iou, float* grad_predict,
                    int B, int n, float divisor);
/* The mesh regularisers of the deformation loop, value and gradient in one launch each (Jittor ops + autograd in the
 * reference).  LaplacianLoss (jrender/loss/laplacian_loss.py:5-37): loss[b] = sum((L x_b)^2), L [NV,NV] given in CSR
 * (rowptr [NV+1], col, val) and, for the gradient 2 L^T L x_b * grad_scale, its transpose in CSR too; scratch
 * [B,NV,3] floats.  FlattenLoss (jrender/loss/flatten_loss.py:5-80): loss[b] = sum over the NE edges shared by two
 * faces of (cos + 1)^2, (v0s, v1s) the edge, (v2s, v3s) the two opposite vertices; gradient * grad_scale (float
 * atomics).  vertices / grad_vertices [B,NV,3]; grad_vertices may be NULL.  Both run a grid per mesh (rows / edges) with a
 * two-stage sum of the loss, so that they scale past one compute unit (a 39k-face mesh has 19 502 vertices). */
int jr_laplacian_loss(jr_ctx* ctx, const int32_t* rowptr, const int32_t* col, const float* val,
                      const int32_t* rowptr_t, const int32_t* col_t, const float* val_t, const float* vertices,
                      float* scratch, float* loss, float* grad_vertices, int B, int NV, float grad_scale);
int jr_flatten_loss(jr_ctx* ctx, const int32_t* v0s, const int32_t* v1s, const int32_t* v2s, const int32_t* v3s,
                    const float* vertices, float* loss, float* grad_vertices, int B, int NV, int NE, float eps,
                    float grad_scale);
/* The optimiser side of the deformation loop (BASELINE configs[3]), Jittor tensor ops + autograd + nn.Adam in the
 * reference, all on the GPU there (demo2-deform.py:17-40, :72, :85-88) and here:
 * jr_deform_vertices_forward   Model.execute, demo2-deform.py:35-41: vertices [NV,3] from the template (already scaled by
 *     0.5, |t| < 1), the displacement map [NV,3] and the centre [3]:
 *       u = sigmoid(log(|t| / (1 - |t|)) + displace) * sign(t);  c = tanh(center);  v = relu(u)(1 - c) - relu(-u)(c + 1) + c
 * jr_deform_vertices_backward  its VJP for the upstream gradient w0*grad0 + w1*grad1 + w2*grad2 (each [NV,3]; grad1 / grad2
 *     may be NULL: the silhouette term and the two regularisers of demo2-deform.py:85-88 are combined here) ->
 *     grad_displace [NV,3], grad_center [3] (a two-stage sum over the vertices: double partial sums per workgroup, added with double
 *     atomics - independent of the order to the float rounding of the result, not bit-deterministic).
 * jr_adam_step   one step of nn.Adam (demo2-deform.py:72) on n floats, in place on param / m / v (m, v start at zero);
 *     step counts from 1; scalars are doubles like the Python floats of the host mirror (jrender_amd/optim.py), which
 *     this kernel reproduces operation by operation:  p -= (lr / (1 - b0^t)) m / (sqrt(v / (1 - b1^t)) + eps).
 * jr_scalar_accumulate   dst[0] = (accumulate ? dst[0] : 0) + bias + scale * sum(src[0..n)): loss terms stay on the
 *     device (e.g. one slot of a history array per iteration) and are read when the caller wants the number. */
int jr_deform_vertices_forward(jr_ctx* ctx, const float* template_vertices, const float* displace, const float* center,
                               float* vertices, int NV);
int jr_deform_vertices_backward(jr_ctx* ctx, const float* template_vertices, const float* displace, const float* center,
                                const float* grad0, float w0, const float* grad1, float w1, const float* grad2, float w2,
                                float* grad_displace, float* grad_center, int NV);
int jr_adam_step(jr_ctx* ctx, float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta0,
                 double beta1, double eps, double weight_decay, int step);
int jr_scalar_accumulate(jr_ctx* ctx, float* dst, const float* src, int n, float scale, float bias, int accumulate);
/* ---- an iteration of an optimisation loop as ONE HIP graph (round 5; the reference's loop - demo2-deform.py:74-90 - issues
 * ~30 small device operations per iteration, here ~0.1 of the 0.55 ms of an iteration is launch overhead) -------------------
 * jr_graph_begin ... jr_graph_end   every call on this context in between is RECORDED (hipStreamBeginCapture on the context's
 *     stream) instead of executed; jr_graph_end returns an executable graph, jr_graph_launch replays it on the context's
 *     stream, jr_graph_destroy frees it, jr_graph_abort closes an open capture after an error.  Rules inside a capture:
 *       - nothing may wait for the GPU or touch host memory (no jr_memcpy_*, jr_synchronize, communicator calls);
 *       - jr_malloc must find a cached block: run the sequence once or twice before capturing it; every block handed out or
 *         freed while the capture is open is PINNED to the graph - once its owner frees it, it is parked (not handed out
 *         again, not released by jr_ctx_trim or the out-of-memory retry) until jr_graph_destroy, because the graph's nodes
 *         address it at every replay; only the rest of the SAME capture may get it again (a graph replays its nodes in the
 *         captured order, so reuse inside one sequence is as safe as it is on the stream).  Buffers allocated BEFORE the capture that the sequence uses stay the caller's to keep
 *         alive for the graph's lifetime;
 *       - the library's own scratch (face records, bin arrays, pair pool, reduction scratch, NMR keys / planes) must not
 *         have to grow inside a capture (the call fails with a message saying so), and a larger call OUTSIDE the graph that
 *         reallocates it outdates every graph captured before: jr_graph_launch then fails instead of replaying onto freed
 *         memory.  jr_graph_launch also fails when the previous replay's forward reported more pairs than the pool holds;
 *       - jr_softras_forward launches its lists and raster kernels against the pool and the launch history of the LAST forward of
 *         the same shape (it cannot wait for the pair total); at every replay the kernels re-check the total on the device and
 *         do nothing when the pool is too small - jr_graph_check (waits for the stream, then compares the last replayed
 *         forward's total with the pool) tells: non-zero = run the sequence outside the graph once and capture again;
 *       - host scalars are frozen into the graph: an iteration number must live on the device -
 * jr_adam_step_counted   jr_adam_step with step = *iteration + 1 read on the device (the bias corrections are formed there, in
 *     double as on the host);
 * jr_scalar_accumulate_at   jr_scalar_accumulate into dst + stride * *iteration (one history row per iteration);
 * jr_counter_add   *counter += delta (one thread): the last node of a captured iteration. */
int jr_graph_begin(jr_ctx* ctx);
int jr_graph_end(jr_ctx* ctx, void** graph_exec);
int jr_graph_abort(jr_ctx* ctx);
int jr_graph_launch(jr_ctx* ctx, void* graph_exec);
int jr_graph_check(jr_ctx* ctx);
int jr_graph_destroy(jr_ctx* ctx, void* graph_exec);
int jr_adam_step_counted(jr_ctx* ctx, float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta0,
                         double beta1, double eps, double weight_decay, const int32_t* iteration);
int jr_scalar_accumulate_at(jr_ctx* ctx, float* dst, int stride, const int32_t* iteration, const float* src, int n, float scale,
                            float bias, int accumulate);
int jr_counter_add(jr_ctx* ctx, int32_t* counter, int delta);
int jr_avgpool2x2_forward(jr_ctx* ctx, const float* in, float* out, int planes, int H, int W);
int jr_avgpool2x2_backward(jr_ctx* ctx, const float* grad_out, float* grad_in, int planes, int H,
                           int W);

/* ---- multi-GPU exchange: RCCL over xGMI, one process per GPU ------------------------------
 * The reference has no distributed code.  A batch shards over ranks with no collective inside the
 * op (every kernel indexes its view independently, SRK:278, :1216); what a caller exchanges
 * afterwards is image / gradient shards (all-gather) and the gradient of vertices shared by all
 * views (all-reduce; demo2-deform.py:45).  Rendezvous: rank 0 calls jr_comm_unique_id and ships
 * the 128 bytes to the other ranks by any side channel (jrender_amd/comm.py: a file); every rank
 * then calls jr_comm_create (collective).  All buffers are DEVICE pointers on the context's GPU,
 * all collectives are enqueued on the context's stream, librccl.so is loaded on first use. */
typedef struct jr_comm jr_comm;
#define JR_COMM_ID_BYTES 128
enum { JR_REDUCE_SUM = 0, JR_REDUCE_MAX = 1 };
int jr_comm_unique_id(void* id_host /* JR_COMM_ID_BYTES */);
int jr_comm_create(jr_ctx* ctx, const void* id_host, int nranks, int rank, jr_comm** out);
int jr_comm_destroy(jr_comm* comm);
int jr_comm_rank(const jr_comm* comm);
int jr_comm_size(const jr_comm* comm);
/* recv[nranks * bytes_per_rank]; in place when send == recv + rank * bytes_per_rank */
int jr_comm_all_gather(jr_comm* comm, const void* send, void* recv, size_t bytes_per_rank);
/* uneven shards: bytes_of_rank is a HOST array [nranks]; recv is their concatenation */
int jr_comm_all_gather_v(jr_comm* comm, const void* send, void* recv, const size_t* bytes_of_rank);
int jr_comm_all_reduce_f32(jr_comm* comm, const float* send, float* recv, size_t count, int op);
/* 1 or 2 HOST doubles reduced over the ranks (timing, loss values); blocks until done */
int jr_comm_all_reduce_host_f64(jr_comm* comm, double* values_host, int count, int op);
int jr_comm_barrier(jr_comm* comm); /* blocks: every rank's stream reached this point */

/* ---- per-phase GPU timing with HIP events on the context stream (benchmarks) ---------
 * After jr_profile_enable(ctx, 1) every forward/backward brackets its phases with event pairs;
 * jr_profile_collect synchronises, returns the summed milliseconds and the number of brackets
 * per phase since the previous collect, and resets. */
enum { JR_PHASE_BIN_COUNT = 0, JR_PHASE_BIN_FILL_SORT = 1, JR_PHASE_FWD_RASTER = 2,
       JR_PHASE_BWD_RASTER = 3, JR_NUM_PHASES = 4 };
int jr_profile_enable(jr_ctx* ctx, int on);
int jr_profile_collect(jr_ctx* ctx, double ms[JR_NUM_PHASES], int64_t launches[JR_NUM_PHASES]);

/* ---- NMR ("n3mr") hard rasteriser: replaces the five JIT ops of
 * jrender/renderer/dr/n3mr/cuda/rasterize.py (forward_face_index_map :5-216, forward_texture_sampling
 * :219-339, backward_pixel_map :342-648, backward_textures :650-727, backward_depth_map :729-821) and
 * the host ops between them (background compositing, alpha = face_index >= 0; n3mr.py:135-148).
 * Layouts are the reference's: faces [B,NF,9]; textures [B,NF,ts,ts,ts,3]; maps are NHWC with
 * BOTTOM-UP rows (the caller flips / permutes like n3mr.py:240-247): face_index_map [B,IS,IS] i32,
 * weight_map [B,IS,IS,3], depth_map [B,IS,IS] (= far where empty), face_inv_map [B,IS,IS,9],
 * rgb_map [B,IS,IS,3] (background already composited), alpha_map [B,IS,IS],
 * sampling_index_map [B,IS,IS,8] i32, sampling_weight_map [B,IS,IS,8], faces_inv [B,NF,9].
 * Pointers of maps that the return_* flags switch off may be NULL.  Depth ties resolve to the lowest
 * face index (deterministic; the reference's spin lock is racy there).  near must be >= 0. */
int jr_n3mr_forward(jr_ctx* ctx, const float* faces, const float* textures, float* faces_inv,
                    int32_t* face_index_map, float* weight_map, float* depth_map, float* face_inv_map,
                    float* rgb_map, float* alpha_map, int32_t* sampling_index_map,
                    float* sampling_weight_map, int B, int NF, int TS, int IS, float near_, float far_,
                    float eps, const float* background_rgb, int return_rgb, int return_alpha,
                    int return_depth);
int jr_n3mr_backward(jr_ctx* ctx, const float* faces, const int32_t* face_index_map,
                     const float* weight_map, const float* depth_map, const float* face_inv_map,
                     const float* rgb_map, const float* alpha_map, const float* sampling_weight_map,
                     const int32_t* sampling_index_map, const float* grad_rgb_map,
                     const float* grad_alpha_map, const float* grad_depth_map, float* grad_faces,
                     float* grad_textures, int B, int NF, int TS, int IS, float eps, int return_rgb,
                     int return_alpha, int return_depth);

/* Output transform of the NMR functional API (n3mr.py:240-256, Jittor tensor ops in the reference): a map in the
 * rasteriser's layout [B,H,W,C] with bottom-up rows -> [B,C,H/pool,W/pool] top-down, 2x2 mean when pool == 2
 * (anti_aliasing); C = 3 for rgb, 1 for alpha / depth.  The backward spreads grad_out / pool^2 back. */
int jr_n3mr_image_forward(jr_ctx* ctx, const float* in_nhwc, float* out_nchw, int B, int H, int W, int C,
                          int pool);
int jr_n3mr_image_backward(jr_ctx* ctx, const float* grad_out_nchw, float* grad_in_nhwc, int B, int H,
                           int W, int C, int pool);

/* ---- self-test of the exact-division identity the kernels rely on (softras_device.h):
 * evaluates n pseudo-random (a, b) pairs on the GPU and counts results of the reciprocal-refinement
 * quotient that differ in any bit from the IEEE quotient a / b.  Must return 0 mismatches. */
int jr_selftest_division(jr_ctx* ctx, uint64_t n, uint32_t seed, uint64_t* mismatches);
/* exhaustive: v_rcp_f32 + one Newton step vs IEEE 1.0f/x for every float with exponent in [-40, 40] */
int jr_selftest_reciprocal(jr_ctx* ctx, uint64_t* mismatches);

/* ---- introspection for tests / benchmarks ---------------------------------------- */
/* statistics of the last forward on this context: [0]=bin-face pairs, [1]=non-empty bins,
 * [2]=max faces in a bin, [3]=bins per image */
int jr_softras_last_stats(jr_ctx* ctx, int64_t stats[4]);
/* which paths the last launches took: [0]=1 when the last forward ran the multi-wavefront kernel (launches of up to 4 Mpixels),
 * [1]=bins whose tiles got a whole workgroup each in it, [2]=wavefronts per workgroup (4 or 8; 1 = one wavefront per tile),
 * [3]=the heavy-bin threshold in force (listed faces) */
int jr_softras_last_launch(jr_ctx* ctx, int64_t info[4]);
/* Launch policy of the multi-wavefront kernels (no counterpart in the reference; results never depend on it, only which
 * kernel organisation computes them).  heavy_min_faces: a bin that lists at least this many faces (rounded DOWN to the
 * launch order's bucket boundary: 12 buckets per octave of the list length, i.e. up to 6 % below the value) gets a whole
 * workgroup per tile in forwards of up to 4 Mpixels (and four wavefronts per tile in small backwards); 0 = never,
 * < 0 = the built-in default of the launch's bin size (512 / 128 or 64 / 48 for 32 / 16 / 8-pixel bins).  heavy_waves: 4 or 8 wavefronts per such workgroup, 0 = chosen per launch from the
 * number of heavy tiles.  Environment JR_FWD_HEAVY_MIN / JR_FWD_HEAVY_WAVES set the same two values at jr_ctx_create. */
int jr_softras_set_launch_policy(jr_ctx* ctx, int heavy_min_faces, int heavy_waves);
/* Screen-bin size in pixels: replaces the `bin_size` argument of the reference's operator (soft_rasterize.py:85-99 ->
 * soft_rasterize_coarse_to_fine.py:16-18; demo2-deform.py:65 passes bin_size=16 for its 64^2 images).  0 = chosen per
 * launch from the image size (the default), otherwise rounded up to the next supported size 8, 16 or 32 (one, 2x2 or 4x4
 * wavefront tiles per bin).  Sticky per context.  It selects how finely the face lists, the launch order and the
 * heavy-bin classification follow the image - results are bit-identical for every value (the reference's own binned path
 * is NOT: it truncates lists at max_elems_per_bin and fills them in a nondeterministic order).  The default threshold of
 * jr_softras_set_launch_policy follows the bin size (512 listed faces for 32-pixel bins, 192 for 16 - 64 for meshes of up to 10 000 faces -, 48 for 8).
 * jr_softras_bin_size: the size a launch of `batch` views of `num_faces` faces at `image_size` would use now (automatic choice:
 * 8-pixel bins up to 128^2 images; 16 up to 512^2 unless the mesh is dense for the image - more than 100 faces per 16-pixel bin
 * on average, num_faces x 256 > 100 x image_size^2 -, which keeps 32; above 512^2 16 while batch x image_size^2 <= 4 Mpixels,
 * else 32; num_faces <= 0: a sparse mesh); image_size <= 0: the size the workspace's current face records / lists were built
 * with (0 before the first forward). */
int jr_softras_set_bin_size(jr_ctx* ctx, int bin_size);
/* Colour-path arithmetic of the forward (sticky per context; default 0).  0: the coverage sigmoid and the softmax weights
 * use the hardware's exp2 / reciprocal (cuda/soft_rasterize.py:338-344, :401-411 evaluate them with expf and a
 * double-precision quotient): RGBA within 5e-5 of the reference, gradients within 1e-4 of the largest component, but
 * 1 - 2e-4 ELEMENT-WISE (|a - b| / (|b| + 1e-3 max|b|)) - the last ulp of D amplified by (k - o) / D / gamma.  1: the
 * reference's own arithmetic for both quantities (a second set of forward kernels): RGBA 8e-6, gradients under 1e-4
 * element-wise; forward +15 % on the headline batch.  The face-index buffer and faces_info are bit-exact in both modes.
 * (Independent of this switch: with euclidean distance and sigma_val < 5e-6 the edge projections of INSIDE pixels keep the
 * reference's IEEE quotients - their reciprocal-multiply form is 1e-7 of an edge length off, which a sigmoid that sharp shows.) */
int jr_softras_set_precise_colour(jr_ctx* ctx, int on);
int jr_softras_bin_size(const jr_ctx* ctx, int image_size, int batch, int num_faces);
/* instrumented builds (-DJR_TUNE_PROFILE_SECTIONS=1, tools/ablate): shader-clock totals per kernel section since the
 * previous call, [0..7] forward raster, [8..15] backward raster; all zero in the product build */
int jr_debug_section_clocks(jr_ctx* ctx, uint64_t clocks[20]);

#ifdef __cplusplus
}
#endif
#endif /* JRENDER_HIP_H */
