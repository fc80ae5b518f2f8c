/*
 * include/xeve_hip.h -- C-ABI of libxeve_hip.so, the MI355X (gfx950) implementation of XEVE's
 * inter-prediction / RDO arithmetic hot path.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *  (1) DROP-IN DISPATCH TABLES with exactly the reference's function-pointer types and index
 *      conventions.  They take borrowed HOST pointers, run the HIP kernel for that one call and
 *      return synchronously -- the literal replacement for the SIMD tables that
 *      xeve_platform_init_func installs (reference: src_base/xeve_enc.c:722-779).  See INTEGRATION.md.
 *
 *  (2) BATCHED DEVICE API: the same arithmetic over many blocks per launch, on picture planes that
 *      are already resident in HBM, asynchronous on a caller-supplied hipStream_t.  This is the form
 *      the GPU is actually fast in (one table call is 128 B..16 KB of work).
 *
 * Error convention: the reference's table functions cannot report failure.  The
 * table layer therefore aborts the process with a message on any HIP error (there is NO CPU
 * fallback, by design); the batched layer returns 0 on success or a negative XEVE_HIP_ERR_* code and
 * keeps a message retrievable with xeve_hip_last_error().
 *
 * Common conventions (reference: src_base/xeve_port.h:54): pel = int16_t, all
 * strides and offsets are in ELEMENTS (not bytes), bit_depth is the codec-internal depth.
 */
#ifndef XEVE_HIP_H
#define XEVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t xeve_hip_pel;

#define XEVE_HIP_OK            0
#define XEVE_HIP_ERR_DEVICE   (-1) /* no usable gfx950 device / HIP runtime error */
#define XEVE_HIP_ERR_ARG      (-2) /* invalid argument */
#define XEVE_HIP_ERR_UNINIT   (-3) /* xeve_hip_init() not called */

/* ------------------------------------------------------------------------------------------- */
/* lifecycle                                                                                   */
/* ------------------------------------------------------------------------------------------- */
/* Binds the calling PROCESS to one GPU (the reference's dispatch globals are process-wide,
 * src_base/xeve_sad.c:34-37, so the binding is too: one encoder process per GPU / GOP shard). */
int         xeve_hip_init(int device_ordinal);
void        xeve_hip_shutdown(void);
const char *xeve_hip_last_error(void);
/* number of table-layer calls served since init (to let tests prove the HIP path ran) */
uint64_t    xeve_hip_table_calls(void);
/* of those, the calls served by the Main-profile entries (xevem_tbl_*_hip, xeve_tbl_tx_hip / _itx_hip) */
uint64_t    xeve_hip_table_calls_main(void);
/* sizeof() of the i-th record type of this header as the library was compiled, in the order xeve_hip_job, _mc_job, _me_params, _me_job, _me_result,
 * _spel_params, _spel_job, _epzs_job, _epzs_params, _sbac, _cu_bits_params, _cu_bits_job, _rdoq_est_full, _deblock_params, _refpic, _cu_mc_job,
 * _rdo_params, _rdo_job, _rdo_result, _skip_job, _skip_result, _inter_params, _inter_job, _inter_result, _intra_params, _intra_job, _intra_result, _tree_params, _ctu_job, _ctu_data, _tree_inter, _eco_params (0 .. 31); -1 past the end.  For bindings in
 * other languages to check their record layouts at load time (no GPU needed). */
int         xeve_hip_sizeof(int i);

/* Kernel-class timers for measurement (bench.py's roofline block): for the classes whose bit is set in class_mask (0 = all off) the batched
 * entry points bracket their launches
 * with HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON and count algorithmic units on the device.  Not for use under stream capture.
 * Classes: 0 integer motion search (k_me_epzs / k_me_diamond; unit = 64 sample pairs of evaluated block SADs, i.e. a w x h evaluation counts
 * w*h/64 units = 256 algorithmic bytes each by SURVEY.md 8d's 4*w*h + 4 per table call), 1 sub-pel stage of the search (fused interpolation
 * + SAD), 2 CABAC bit counting (k_cu_bits; unit = one coded bin), 3 CU prediction (xeve_hip_mc_cu_jobs), 4 residual chain (DIFF .. SSD),
 * 5 RDOQ; classes 1, 3, 4, 5 are timed only (units stay 0); 6 is a pure counter: bit-count jobs that could not use their blocks' bin strings
 * (string overflow or an inconsistent coefficient count) and took the slower event automaton.  xeve_hip_prof_read waits for the recorded events, adds up their durations per class and resets the tallies;
 * arrays of n <= 8 entries (NULL to skip one). */
#define XEVE_HIP_PROF_CLASSES 8
int         xeve_hip_prof_enable(int class_mask);
int         xeve_hip_prof_read(double *ms, uint64_t *launches, uint64_t *units, int n);
/* bins-per-job histogram of the CABAC bit-count launches that ran with their class timer on: out[b] (32 entries) = jobs whose bin count has bit length b (0: no bin, 1: one,
 * 2: 2-3, 3: 4-7, ...); reset != 0 clears it.  Measurement only. */
int         xeve_hip_prof_cu_bits_hist(unsigned long long *out, int reset);

/* ------------------------------------------------------------------------------------------- */
/* (1) drop-in dispatch tables                                                                 */
/* ------------------------------------------------------------------------------------------- */
/* reference: src_base/xeve_sad.h:41-45 */
typedef int     (*XEVE_HIP_FN_SAD)(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth);
typedef int     (*XEVE_HIP_FN_SATD)(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth);
typedef int64_t (*XEVE_HIP_FN_SSD)(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth);
typedef void    (*XEVE_HIP_FN_DIFF)(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int s_diff,
                                    int16_t *diff, int bit_depth);
/* reference: src_base/xeve_mc.h:85-87 */
typedef void (*XEVE_HIP_MC_L)(xeve_hip_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, xeve_hip_pel *pred,
                              int w, int h, int bit_depth, const int16_t (*mc_l_coeff)[8]);
typedef void (*XEVE_HIP_MC_C)(xeve_hip_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, xeve_hip_pel *pred,
                              int w, int h, int bit_depth, const int16_t (*mc_c_coeff)[4]);
typedef void (*XEVE_HIP_AVG_NO_CLIP)(int16_t *src, int16_t *ref, int16_t *dst, int s_src, int s_ref, int s_dst,
                                     int wd, int ht);
/* reference: src_base/xeve_type.h:169-170 */
typedef void (*XEVE_HIP_TXB)(void *coef, void *t, int shift, int line, int step);
typedef void (*XEVE_HIP_ITXB)(void *coef, void *t, int shift, int line, int step);

/* replaces xeve_tbl_sad_16b{,_sse,_avx}   (xeve_sad.c:66, sse/xeve_sad_sse.c:254, avx/xeve_sad_avx.c:108); [log2 w][log2 h] */
extern const XEVE_HIP_FN_SAD  xeve_tbl_sad_16b_hip[8][8];
/* replaces xeve_tbl_ssd_16b{,_sse}        (xeve_sad.c:300, sse/xeve_sad_sse.c:1003) */
extern const XEVE_HIP_FN_SSD  xeve_tbl_ssd_16b_hip[8][8];
/* replaces xeve_tbl_diff_16b{,_sse}       (xeve_sad.c:183, sse/xeve_sad_sse.c:523) */
extern const XEVE_HIP_FN_DIFF xeve_tbl_diff_16b_hip[8][8];
/* replaces xeve_tbl_satd_16b{,_sse}       (xeve_sad.c:1143, sse/xeve_sad_sse.c:2722) */
extern const XEVE_HIP_FN_SATD xeve_tbl_satd_16b_hip[1];
/* replaces xeve_tbl_mc_l/_c{,_sse,_avx}   (xeve_mc.c:383-399); [dx != 0][dy != 0] */
extern const XEVE_HIP_MC_L    xeve_tbl_mc_l_hip[2][2];
extern const XEVE_HIP_MC_C    xeve_tbl_mc_c_hip[2][2];
/* replaces xeve_average_16b_no_clip{,_sse} (xeve_mc.c:449) */
void xeve_average_16b_no_clip_hip(int16_t *src, int16_t *ref, int16_t *dst, int s_src, int s_ref, int s_dst, int wd, int ht);
/* replaces xeve_tbl_txb{,_avx} (xeve_tq.c:394) and xeve_tbl_itxb{,_sse,_avx} (xeve_itdq.c:432); [log2 N - 1];
 * these two are installed BY ADDRESS (&table), like the reference's (xeve_enc.c:752-753) */
extern const XEVE_HIP_TXB     xeve_tbl_txb_hip[6];
extern const XEVE_HIP_ITXB    xeve_tbl_itxb_hip[6];
/* replaces xeve_recon_blk (xeve_recon.c:34; installed as ctx->fn_recon, xeve_enc.c:822) */
void xeve_recon_blk_hip(int16_t *coef, xeve_hip_pel *pred, int is_coef, int cuw, int cuh, int s_rec, xeve_hip_pel *rec, int bit_depth);

/* ---- Main profile, first slice: the entries the Main tools add to the dispatch layer ---- */
/* reference: src_main/xevem_mc.h:45 (XEVEM_MC: like XEVE_MC_L without the coefficient argument -- the Main filters are fixed) */
typedef void (*XEVE_HIP_MCM)(xeve_hip_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, xeve_hip_pel *pred, int w, int h, int bit_depth);
/* reference: XEVE_TX / XEVE_ITX, the 16-bit-intermediate 1-D transforms of tool_iqt (src_main/xevem_tq.c:58, xevem_itdq.c:302) */
typedef void (*XEVE_HIP_TX)(int16_t *coef, int16_t *t, int shift, int line);
/* replace xevem_tbl_dmvr_mc_l / _c{,_sse} and xevem_tbl_bl_mc_l{,_sse} (xevem_mc.c:465-485); [dx != 0][dy != 0].
 * DMVR: `ref` points AT the block, only the fraction of gmv counts; bilinear: gmv's integer part moves ref, footprint (w+1) x (h+1). */
extern const XEVE_HIP_MCM     xevem_tbl_dmvr_mc_l_hip[2][2];
extern const XEVE_HIP_MCM     xevem_tbl_dmvr_mc_c_hip[2][2];
extern const XEVE_HIP_MCM     xevem_tbl_bl_mc_l_hip[2][2];
/* replace xeve_tbl_tx{,_avx} (xevem_tq.c:702) and xeve_tbl_itx{,_avx} (xevem_itdq.c:549); [log2 N - 1]; shift >= 1 for N >= 4
 * (the reference forms 1 << (shift - 1) unguarded there); installed BY ADDRESS like xeve_func_txb (xevem_util.c:3948-3963) */
extern const XEVE_HIP_TX      xeve_tbl_tx_hip[6];
extern const XEVE_HIP_TX      xeve_tbl_itx_hip[6];
/* reference: XEVE_INTRA_PRED_ANG (src_main/xevem_ipred.h:104-112); src_* point at index 0 of neighbour lines indexed -1 .. w + h - 1 */
typedef void (*XEVE_HIP_INTRA_PRED_ANG)(xeve_hip_pel *src_le, xeve_hip_pel *src_up, xeve_hip_pel *src_ri, uint16_t avail_lr, xeve_hip_pel *dst, int w, int h, int ipm,
                                        int bit_depth);
/* replaces xeve_tbl_intra_pred_ang (xevem_ipred.c:811-815): [ipm < IPD_VER | ipm > IPD_HOR | between][right line used]; ipm 3 .. 32 except 12 and 24 */
extern const XEVE_HIP_INTRA_PRED_ANG xeve_tbl_intra_pred_ang_hip[3][2];
/* reference: XEVE_INV_TRANS (src_main/xevem_type.h:47): (coef, block, shift, line, skip_line, skip_line_2) */
typedef void (*XEVE_HIP_INV_TRANS)(int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2);
/* replaces xeve_itrans_map_tbl{,_sse} (xevem_itdq.c:42-47): [0] DCT-VIII, [1] DST-VII; [.][log2 N - 1], N = 4 .. 32 ([.][0] and rows 2 .. 15 are NULL there too) */
extern const XEVE_HIP_INV_TRANS xeve_itrans_map_tbl_hip[16][5];
/* the forward passes: the entries of xeve_trans_map_tbl (xevem_tq.c:53-56; `Trans`, :40-41: (block, coef, shift, line, skip_line, skip_line_2), the same argument list with
 * input and output exchanged) -- an array the reference indexes directly (xeve_t_MxN_ats_intra, :698-699), so the binding assigns its entries:
 * xeve_trans_map_tbl[t][n] = xeve_trans_map_tbl_hip[t][n] for t < 2, n = 1 .. 4 */
extern const XEVE_HIP_INV_TRANS xeve_trans_map_tbl_hip[16][5];
/* replace xevem_scaled_horizontal / _vertical_sobel_filter{,_sse} and xevem_equal_coeff_computer{,_sse} (xevem_mc.c:2341-2447): the kernels of the affine
 * gradient search; 3 <= width, height <= 128; equal_coeff is accumulated into (the caller zeroes it); residue is read with derivate_buf_stride, as the reference does */
void xevem_scaled_horizontal_sobel_filter_hip(xeve_hip_pel *pred, int pred_stride, int *derivate, int derivate_buf_stride, int width, int height);
void xevem_scaled_vertical_sobel_filter_hip(xeve_hip_pel *pred, int pred_stride, int *derivate, int derivate_buf_stride, int width, int height);
void xevem_equal_coeff_computer_hip(xeve_hip_pel *residue, int residue_stride, int **derivate, int derivate_buf_stride, int64_t (*equal_coeff)[7], int width,
                                    int height, int vertex_num);
/* Zero-edit installation into a loaded MAIN-profile reference library: everything xeve_hip_install_tables patches (the Main
 * library carries the same Baseline globals) plus xevem_func_dmvr_mc_l / _c, xevem_func_bl_mc_l (xevem_mc.c:39-41), xeve_func_tx
 * (xevem_tq.c:41), xeve_func_itx (xevem_itdq.c:39), xeve_func_itrans (xevem_itdq.c:51) and xevem_func_aff_h_sobel_flt / _v_sobel_flt / _eq_coef_comp
 * (xevem_mc.c:42-44) and xeve_func_intra_pred_ang (xevem_ipred.c:38) -- every entry of xevem_platform_init_func.  Returns the number of pointers patched
 * (8 + 10 [+ 1]) or a negative error. */
int xeve_hip_install_tables_main(void *fn_itxb_slot);

/* Zero-edit installation: overwrites the reference library's exported pointer globals
 * (xeve_func_sad/ssd/diff/satd, xeve_func_mc_l/mc_c, xeve_func_average_no_clip, xeve_func_txb --
 * xeve_sad.c:34-37, xeve_mc.c:34-36, xeve_tq.c:36) found with dlsym(RTLD_DEFAULT) in the calling
 * process.  `fn_itxb_slot` is the address of ctx->fn_itxb (xeve_type.h:984) or NULL.  Returns the number
 * of pointers patched (8 + 1), or a negative error when the reference library is not loaded. */
int xeve_hip_install_tables(void *fn_itxb_slot);

/* ------------------------------------------------------------------------------------------- */
/* (2) batched device API -- every pointer below is a DEVICE pointer unless stated; `stream` is   */
/*     a hipStream_t (NULL = default stream); calls are asynchronous on that stream.             */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_job {
    int32_t off1; /* element offset of the block's top-left sample inside plane 1 (e.g. the original): never negative, and read as an UNSIGNED 32-bit number (the stacked
                     originals of a picture batch span up to 2^32 samples) */
    int32_t off2; /* element offset inside plane 2 (e.g. the reference picture at the search centre), below 2^30 or negative.  (A non-negative off2 with bit 30 set marks
                     a record the library built for itself: off1 counts PAIRS of samples there -- 2^33 samples, one batch over all of a GPU's HBM -- and the low 30 bits of
                     off2 are the offset; a caller's records never need it.) */
} xeve_hip_job;

#define XEVE_HIP_SRC1_SIGNED 1 /* plane 1 may hold negative samples (org_bi = 2*org - pred, xeve_pinter.c:143-156) */

/* out[j * ncand + c] = sad(w, h, p1 + jobs[j].off1, p2 + jobs[j].off2 + cand_off[c], s1, s2, bit_depth)
 * (reference semantics: sad_16b, xeve_sad.c:40-61).  cand_off = element offsets (dy * s2 + dx) of the
 * candidates of one search round relative to the job's centre; ncand >= 1. */
int xeve_hip_sad_jobs(const xeve_hip_pel *p1, int s1, const xeve_hip_pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                      const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int flags, int32_t *out, void *stream);
/* Alignment-optimised form.  On gfx950 a vector load from a non-dword-aligned address (odd pel position) runs at
 * about a third of the aligned rate, and half of all search candidates sit at odd x.  The caller keeps, next to each
 * reference plane, a copy shifted by one element (p2_shift1[i] == p2[i + 1], made once per picture with
 * xeve_hip_plane_shift1); odd positions are then read from the copy at an even address.  Same results. */
int xeve_hip_plane_shift1(const xeve_hip_pel *src, xeve_hip_pel *dst, int64_t n_elements, void *stream);
int xeve_hip_sad_jobs_dual(const xeve_hip_pel *p1, int s1, const xeve_hip_pel *p2, const xeve_hip_pel *p2_shift1, int s2,
                           const xeve_hip_job *jobs, int njobs, const int32_t *cand_off, int ncand, int w, int h, int bit_depth,
                           int flags, int32_t *out, void *stream);
/* same job structure; ssd_16b (xeve_sad.c:275-297) and xeve_had (xeve_sad.c:1043-1140) */
int xeve_hip_ssd_jobs(const xeve_hip_pel *p1, int s1, const xeve_hip_pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                      const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int64_t *out, void *stream);
int xeve_hip_satd_jobs(const xeve_hip_pel *p1, int s1, const xeve_hip_pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                       const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int32_t *out, void *stream);
/* diff[j][y][x] (dense w*h per job) = p1[..] - p2[..]   (diff_16b, xeve_sad.c:160-178) */
int xeve_hip_diff_jobs(const xeve_hip_pel *p1, int s1, const xeve_hip_pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                       int w, int h, int16_t *diff, void *stream);

typedef struct xeve_hip_mc_job {
    int32_t gmv_x, gmv_y; /* absolute position relative to `ref`, 1/16 pel luma, 1/32 pel chroma (xeve_mc.c:481-488) */
    int32_t pred_off;     /* element offset of the output block inside `pred` */
    int32_t frac;         /* bit0: horizontal filter, bit1: vertical filter -- the table index [dx!=0][dy!=0],
                             which the reference derives from the UNCLIPPED mv (xeve_mc.h:96-104) */
} xeve_hip_mc_job;
/* xeve_mc_l_{00,n0,0n,nn} (xeve_mc.c:99-254); coef is a HOST pointer to the [16][8] table in use */
int xeve_hip_mc_l_jobs(const xeve_hip_pel *ref, int s_ref, xeve_hip_pel *pred, int s_pred, const xeve_hip_mc_job *jobs, int njobs,
                       int w, int h, int bit_depth, const int16_t (*coef)[8], void *stream);
/* xeve_mc_c_{00,n0,0n,nn} (xeve_mc.c:259-381); coef is a HOST pointer to the [32][4] table in use */
int xeve_hip_mc_c_jobs(const xeve_hip_pel *ref, int s_ref, xeve_hip_pel *pred, int s_pred, const xeve_hip_mc_job *jobs, int njobs,
                       int w, int h, int bit_depth, const int16_t (*coef)[4], void *stream);
/* Fused forms: the interpolated block is compared with the original at org + jobs[j].pred_off (pred_off is reused as the
 * block's offset inside `org`) and never written.  sad[j] as xeve_mc_l + xeve_sad_16b in me_spel_pattern
 * (xeve_pinter.c:593-627); ssd[j] as the MC + xeve_ssd_16b of the skip/merge analysis (xeve_pinter.c:1437-1458).
 * luma: w % 8 == 0; chroma: w % 4 == 0. */
int xeve_hip_mc_l_sad_jobs(const xeve_hip_pel *ref, int s_ref, const xeve_hip_pel *org, int s_org, const xeve_hip_mc_job *jobs, int njobs,
                           int w, int h, int bit_depth, const int16_t (*coef)[8], int32_t *sad, void *stream);
int xeve_hip_mc_ssd_jobs(int luma, const xeve_hip_pel *ref, int s_ref, const xeve_hip_pel *org, int s_org, const xeve_hip_mc_job *jobs,
                         int njobs, int w, int h, int bit_depth, const void *coef /* [16][8] luma or [32][4] chroma */, int64_t *ssd,
                         void *stream);
/* dst = (a + b + 1) >> 1 over n dense samples (xeve_average_16b_no_clip, xeve_mc.c:449-463) */
int xeve_hip_avg(const int16_t *a, const int16_t *b, int16_t *dst, int64_t n, void *stream);

/* 2-D forward transform of nblk dense blocks, in place: xeve_trans (xeve_tq.c:396-404) */
int xeve_hip_trans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream);
/* 2-D inverse transform, in place: xeve_itrans (xeve_itdq.c:435-440) */
int xeve_hip_itrans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream);
/* plain quantisation (xeve_quant_nnz, rdoq == 0 branch, xeve_tq.c:704-727); nnz[b] = non-zero count; may be NULL */
int xeve_hip_quant(int16_t *coef, int nblk, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth,
                   int32_t *nnz, void *stream);
/* RDOQ all-zero pre-test (xeve_tq.c:666-699): coded[b] = 1 if the block survives, else 0 and the block is zeroed */
int xeve_hip_rdoq_zero_test(int16_t *coef, int nblk, int log2w, int log2h, int qp, int scale, int is_intra_slice,
                            int bit_depth, int32_t *coded, void *stream);
/* RDOQ: xeve_rdoq_run_length_cc (xeve_tq.c:497-649) over nblk dense blocks, in place; nnz[b] = its return value.
 * The reference walks the zig-zag scan sequentially (the rate of a coefficient depends on whether the previous one in
 * scan order was quantised to zero, the "last" position on a running cost).  In Baseline the context index is constant
 * per component (xeve_rdoq_set_ctx_cc, xeve_tq.c:492-495), so the dependency is a two-state automaton: the kernel
 * evaluates both outcomes per coefficient and resolves states, running costs and the best last position with parallel
 * prefix scans -- same levels, same nnz.  `est` (HOST pointer) holds the CABAC-derived bit estimates the reference keeps
 * in XEVE_CORE (xeve_type.h:737-747): cbf = the pair chosen for this component / slice type (xeve_tq.c:565-583).
 * lambda as the reference receives it (double); tool_iqt selects the row of xeve_quant_scale (0 in Baseline). */
typedef struct xeve_hip_rdoq_est {
    int32_t cbf[2];
    int32_t run[24][2], level[24][2], last[2][2];
} xeve_hip_rdoq_est;
int xeve_hip_rdoq(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                  const xeve_hip_rdoq_est *est, int32_t *nnz, void *stream);
/* xeve_quant_nnz with use_rdoq = 1 (xeve_tq.c:651-703): the all-zero pre-test (zero_test != 0), then RDOQ */
int xeve_hip_rdoq_zt(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                     const xeve_hip_rdoq_est *est, int zero_test, int is_intra_slice, int32_t *nnz, void *stream);
/* The estimates as the reference derives them: xeve_rdoq_bit_est (xeve_mode.c:326-372) turns a coder state into
 * core->rdoq_est_* (xeve_type.h:737-747) through the table of xeve_init_bits_est (xeve_mode.c:304-313).  One record per
 * state, device memory; xeve_hip_rdoq_dev then quantises block b with record est[est_idx[b]] (est_idx == NULL: record 0)
 * and picks the cbf pair like xeve_tq.c:565-583 (ch_type 0 Y / 1 U / 2 V; is_intra_cu = the CU's own mode, while the zero
 * pre-test's threshold follows the SLICE type, xeve_tq.c:684-686). */
struct xeve_hip_sbac;
typedef struct xeve_hip_rdoq_est_full {
    int32_t cbf_all[2], cbf_luma[2], cbf_cb[2], cbf_cr[2];
    int32_t run[24][2], level[24][2], last[2][2];
} xeve_hip_rdoq_est_full;
int xeve_hip_rdoq_bit_est(const struct xeve_hip_sbac *sbac, int nstates, xeve_hip_rdoq_est_full *est, void *stream);
int xeve_hip_rdoq_dev(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int ch_type, int bit_depth, int tool_iqt,
                      const xeve_hip_rdoq_est_full *est, const int32_t *est_idx, int zero_test, int is_intra_slice, int is_intra_cu,
                      int32_t *nnz, void *stream);
/* One transform block in HOST memory, synchronous (the table layer's style): xeve_tq_nnz (xeve_tq.c:729-748: xeve_trans + xeve_quant_nnz with
 * the zero pre-test and RDOQ, or the plain quantiser) and itdq_cu (xeve_itdq.c:454-497: xeve_dquant + xeve_itrans) -- what the per-component loops
 * of ctx->fn_tq (xeve_sub_block_tq) and ctx->fn_itdp (xeve_itdq) call.  est: HOST record holding core->rdoq_est_* (may be NULL when !use_rdoq). */
int xeve_hip_tq_nnz_host(int16_t *coef, int log2w, int log2h, int qp, double lambda, int ch_type, int is_intra_cu, int is_intra_slice, int bit_depth,
                         int tool_iqt, const xeve_hip_rdoq_est_full *est, int use_rdoq, int32_t *nnz);
int xeve_hip_itdq_host(int16_t *coef, int log2w, int log2h, int qp, int bit_depth);
/* xeve_dquant with itdq_cu's shift/offset (xeve_itdq.c:442-475) */
int xeve_hip_dquant(int16_t *coef, int nblk, int log2w, int log2h, int scale, int bit_depth, void *stream);
/* xeve_recon_blk over nblk dense blocks; rec block b is written at rec + rec_off[b] with stride s_rec;
 * is_coef[b] as in the reference (xeve_recon.c:34-57); is_coef == NULL means all 1 */
int xeve_hip_recon(const int16_t *coef, const xeve_hip_pel *pred, const uint8_t *is_coef, int nblk, int cuw, int cuh,
                   const int32_t *rec_off, int s_rec, xeve_hip_pel *rec, int bit_depth, void *stream);

/* Fused residual chain of one inter CU component -- the arithmetic core of pinter_residue_rdo
 * (src_base/xeve_pinter.c:962-1051) in ONE launch, with the block held on-chip between the steps:
 *   resi = org - pred (xeve_diff_pred, xeve_mode.c:2737)      ssd[j][0] = SSD(org, pred)         (xeve_pinter.c:984)
 *   coef = quant(DCT(resi))  (xeve_trans + RDOQ zero pre-test + plain quant, xeve_tq.c:396-404,666-727)
 *   resi' = IDCT(dequant(coef))                                (itdq_cu, xeve_itdq.c:454-497)
 *   rec = clip(resi' + pred)  (xeve_recon_blk)                 ssd[j][1] = SSD(org, rec)          (xeve_pinter.c:1037-1051)
 * jobs[j].off1 = block position in `org` AND in `rec` (two planes of identical geometry, strides s_org / s_rec);
 * jobs[j].off2 = position of the prediction block inside `pred` (stride s_pred; a dense [njobs][h*w] buffer has
 * s_pred = w, off2 = j*w*h).  coef receives the quantised levels ([njobs][h*w]); nnz[j] their count.
 * zero_test != 0 applies the RDOQ all-zero pre-test before quantising, as the presets with rdoq = 1 do. */
int xeve_hip_residual_rdo(const xeve_hip_pel *org, int s_org, const xeve_hip_pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs,
                          int log2w, int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, int zero_test,
                          int16_t *coef, xeve_hip_pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, void *stream);
/* The same chain with the quantiser the presets configure (rdoq = 1): zero pre-test + xeve_rdoq_run_length_cc between the
 * forward and the inverse half (three launches: front half, RDOQ scan kernel, back half).  est / lambda / is_luma /
 * tool_iqt as for xeve_hip_rdoq. */
int xeve_hip_residual_rdoq(const xeve_hip_pel *org, int s_org, const xeve_hip_pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs,
                           int log2w, int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, double lambda,
                           int is_luma, int tool_iqt, const xeve_hip_rdoq_est *est, int16_t *coef, xeve_hip_pel *rec, int s_rec,
                           int32_t *nnz, int64_t *ssd, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (3) GPU-side consumer of the SAD kernel: one complete me_ipel_diamond per job (SURVEY.md 8(f)   */
/*     rank 2).  reference: src_base/xeve_pinter.c:363-551 (+ get_mv_bits :74-120, MV_COST :47,    */
/*     get_range_ipel :122-140).  Bit-exact incl. the tie-break (first strictly smaller cost in   */
/*     evaluation order).  One wave per job; per-candidate sums and the per-round minimum are     */
/*     wave64 cross-lane reductions.                                                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_me_params {
    uint32_t lambda_mv;        /* pi->lambda_mv */
    int32_t  refi_bits;        /* xeve_tbl_refi_bits[num_refp][refi] */
    int32_t  extra_bits;       /* bi ? pi->mot_bits[other list] : 0 */
    int32_t  bi;               /* 0 BI_NON; 1 BI_NORMAL (+-5 grid, one round); 2/3 BI_FL0/BI_FL1 (xeve_def.h:504-507) */
    int32_t  faststep;         /* MAX_FIRST_SEARCH_STEP / MAX_REFINE_SEARCH_STEP (xeve_pred.h:63-69) */
    int32_t  max_search_range; /* pi->max_search_range */
    int32_t  range_recentre;   /* range get_range_ipel derives for this reference picture (POC-distance scaled) */
    int32_t  min_clip[2], max_clip[2]; /* pi->min_clip / pi->max_clip */
    int32_t  reserved;         /* xeve_hip_me_epzs_jobs only: bit 0 = me_raster on (pi->me_complexity > 1), bits 8..15 = refi (its step scales with refi + 1) */
} xeve_hip_me_params;
typedef struct xeve_hip_me_job {
    int32_t x, y;     /* block position (integer pel, picture coordinates) */
    int32_t org_off;  /* bi != 0: element offset of the job's dense org_bi block (stride w) inside `org_bi` */
    int16_t range[4]; /* min x, min y, max x, max y */
    int16_t gmvp[2];  /* MVP, picture coordinates, quarter pel */
    int16_t mvi[2];   /* initial MV, picture coordinates, quarter pel */
    int32_t beststep_in; /* *beststep on entry (the reference threads `tmpstep` through successive calls) */
} xeve_hip_me_job;
typedef struct xeve_hip_me_result {
    int16_t  mv[2];    /* best MV relative to the block (quarter-pel units) */
    uint32_t cost;     /* cost_best (MV_COST + SAD) */
    int32_t  beststep; /* *beststep on exit */
    int32_t  best_mv_bits;
} xeve_hip_me_result;
/* org0 / ref0 point at sample (0,0) of the picture inside its padded plane (device memory); org_bi may be NULL when
 * params->bi == 0; params is a HOST pointer; square blocks 8..64. */
int xeve_hip_me_ipel_diamond_jobs(const xeve_hip_pel *org0, int s_org, const xeve_hip_pel *org_bi, const xeve_hip_pel *ref0, int s_ref,
                                  const xeve_hip_me_job *jobs, int njobs, int log2w, int log2h, int bit_depth,
                                  const xeve_hip_me_params *params, xeve_hip_me_result *results, void *stream);

/* Sub-pel refinement: one complete me_spel_pattern per job (src_base/xeve_pinter.c:553-697): `hpel_cnt` half-pel points
 * around the integer result (xeve_pinter.c:67-70), then -- when qpel_cnt > 0, i.e. me_level > ME_LEV_HPEL -- `qpel_cnt`
 * quarter-pel points around the half-pel winner (xeve_pinter.c:50-55); each candidate = xeve_mc_l + xeve_sad_16b (fused,
 * k_mc<OUT=1>) + MV_COST, first strictly smaller cost wins.  results[j].best_mv_bits follows the reference (only the
 * quarter-pel stage updates it); results[j].beststep is 0. */
typedef struct xeve_hip_spel_params {
    uint32_t lambda_mv;
    int32_t  refi_bits, extra_bits, bi; /* as in xeve_hip_me_params */
    int32_t  hpel_cnt, qpel_cnt;        /* pi->search_pattern_hpel_cnt; pi->search_pattern_qpel_cnt or 0 */
} xeve_hip_spel_params;
typedef struct xeve_hip_spel_job {
    int32_t x, y;     /* block position (integer pel) */
    int32_t org_off;  /* bi != 0: offset of the job's dense org_bi block */
    int16_t gmvp[2];  /* MVP, picture coordinates, quarter pel */
    int16_t mvi[2];   /* starting MV relative to the block, quarter pel (the integer search's result) */
} xeve_hip_spel_job;
/* workspace: device scratch of at least xeve_hip_me_spel_workspace(njobs) bytes; coef: HOST pointer to the [16][8] table */
size_t xeve_hip_me_spel_workspace(int njobs);
int xeve_hip_me_spel_pattern_jobs(const xeve_hip_pel *org0, int s_org, const xeve_hip_pel *org_bi, const xeve_hip_pel *ref0, int s_ref,
                                  const xeve_hip_spel_job *jobs, int njobs, int log2w, int log2h, int bit_depth,
                                  const int16_t (*coef)[8], const xeve_hip_spel_params *params, xeve_hip_me_result *results,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* The whole per-list search of one block: pinter_me_epzs (src_base/xeve_pinter.c:699-869) for me_complexity == 1 (no
 * raster search) and me_level > ME_LEV_IPEL -- first diamond search from the MVP (or, bi == 1, from mv_start), refinement
 * diamond searches from the running best while beststep > 0, then the sub-pel pattern search.  The integer stage is ONE
 * kernel (a wave per block loops over its searches, the original block in registers throughout); the call is asynchronous
 * on `stream` like the rest of the batched API.  results[j].mv / .cost are what pinter_me_epzs returns; .best_mv_bits is what the searches leave in
 * pi->mot_bits[lidx] (0: they leave it untouched; xeve_pinter.c:546-548,690-692); .beststep is 0.  A job with x < 0 is switched off (no work;
 * its result is unspecified).  The other branches of the function: params->me.reserved bit 0 adds me_raster (xeve_pinter.c:158-268) after a first
 * search that ended with beststep > 5 (me_complexity > 1: preset placebo); params->hpel_cnt == 0 replaces the sub-pel stage by me_ipel_refinement
 * (xeve_pinter.c:270-361; me_level = ME_LEV_IPEL). */
typedef struct xeve_hip_epzs_job {
    int32_t x, y;        /* block position (integer pel) */
    int32_t org_off;     /* bi != 0: offset of the job's dense org_bi block */
    int16_t mvp[2];      /* MV predictor relative to the block, quarter pel */
    int16_t mv_start[2]; /* bi == 1 only: the MV to refine */
} xeve_hip_epzs_job;
typedef struct xeve_hip_epzs_params {
    xeve_hip_me_params me; /* faststep is ignored: 3 for the first search, 2 for refinements (xeve_pred.h:64-65) */
    int32_t hpel_cnt, qpel_cnt;
} xeve_hip_epzs_params;
size_t xeve_hip_me_epzs_workspace(int njobs);
int xeve_hip_me_epzs_jobs(const xeve_hip_pel *org0, int s_org, const xeve_hip_pel *org_bi, const xeve_hip_pel *ref0, int s_ref,
                          const xeve_hip_epzs_job *jobs, int njobs, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                          const xeve_hip_epzs_params *params, xeve_hip_me_result *results, void *workspace, size_t workspace_bytes,
                          void *stream);

/* The same with pi->mot_bits[other list] per job (bi == 1: a batch of bi-prediction searches whose CUs searched the other list with
 * different outcomes): extra_bits[j] replaces params->me.extra_bits; device memory, NULL = the common value. */
int xeve_hip_me_epzs_jobs_x(const xeve_hip_pel *org0, int s_org, const xeve_hip_pel *org_bi, const xeve_hip_pel *ref0, int s_ref,
                            const xeve_hip_epzs_job *jobs, int njobs, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                            const xeve_hip_epzs_params *params, const int32_t *extra_bits, xeve_hip_me_result *results, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (4) CABAC (SBAC) bit counting of an inter CU -- the rate term of pinter_residue_rdo and of   */
/*     the skip / merge analysis.  reference: src_base/xeve_mode.c:39-295 */
/*     (xeve_sbac_bit_reset, xeve_get_bit_number, xeve_rdo_bit_cnt_cu_inter / _cu_inter_comp /   */
/*     _cu_skip) over src_base/xeve_eco.c (xeve_sbac_encode_bin :521-575, sbac_encode_bin_ep     */
/*     :455-472, sbac_carry_propagate :429-453, xeve_eco_run_length_cc :707-771, xeve_eco_cbf    */
/*     :793-894, xeve_eco_mvd :1235-1280, xeve_eco_refi :1158-1188, xeve_eco_mvp_idx :1190-1203, */
/*     xeve_eco_inter_pred_idc :1123-1156).  Baseline tool set: tool_admvp 0, no delta QP, CU <=  */
/*     64x64 (one transform block per component).  The arithmetic coder is inherently serial per  */
/*     CU, so the unit of parallelism is the job: one lane per job, every lane stepping the same  */
/*     one-bin-per-iteration state machine.                                                       */
/* ------------------------------------------------------------------------------------------- */
/* the fields of XEVE_SBAC (xeve_type.h:527-540) and the context models of XEVE_SBAC_CTX (xeve_def.h:736-790) the
 * inter-CU syntax touches: ctx[XEVE_HIP_CTX_x + i] = sbac->ctx.x[i] */
enum {
    XEVE_HIP_CTX_SKIP_FLAG = 0,  /* [2]  */
    XEVE_HIP_CTX_PRED_MODE = 2,  /* [3]  */
    XEVE_HIP_CTX_DIRECT    = 5,  /* [1]  direct_mode_flag */
    XEVE_HIP_CTX_INTER_DIR = 6,  /* [2]  */
    XEVE_HIP_CTX_REFI      = 8,  /* [2]  */
    XEVE_HIP_CTX_MVP_IDX   = 10, /* [3]  */
    XEVE_HIP_CTX_MVD       = 13, /* [1]  */
    XEVE_HIP_CTX_CBF_ALL   = 14, XEVE_HIP_CTX_CBF_LUMA = 15, XEVE_HIP_CTX_CBF_CB = 16, XEVE_HIP_CTX_CBF_CR = 17,
    XEVE_HIP_CTX_RUN       = 18, /* [24] */
    XEVE_HIP_CTX_LAST      = 42, /* [2]  */
    XEVE_HIP_CTX_LEVEL     = 44, /* [24] */
    XEVE_HIP_CTX_INTRA_DIR = 68, /* [2] sbac->ctx.intra_dir (the Baseline intra prediction mode)  */
    XEVE_HIP_CTX_SPLIT_CU  = 70, /* [1] sbac->ctx.split_cu_flag                                     */
    XEVE_HIP_CTX_DELTA_QP  = 71, /* [1] sbac->ctx.delta_qp                                          */
    XEVE_HIP_SBAC_NCTX     = 72
};
typedef struct xeve_hip_sbac {
    uint32_t range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter;
    uint16_t ctx[XEVE_HIP_SBAC_NCTX];
} xeve_hip_sbac;
typedef struct xeve_hip_cu_bits_params {
    int32_t log2_cuw, log2_cuh;  /* 2..6 */
    int32_t slice_type;          /* XEVE_ST_B 0 / XEVE_ST_P 1 / XEVE_ST_I 2 (inc/xeve.h:170-172) */
    int32_t num_refp[2];         /* ctx->rpm.num_refp */
    int32_t cm_init;             /* sps_cm_init_flag (0 in Baseline) */
    int32_t chroma_format_idc;   /* 0..3; chroma block = (w >> w_shift) x (h >> h_shift), XEVE_GET_CHROMA_{W,H}_SHIFT */
} xeve_hip_cu_bits_params;
enum { XEVE_HIP_BITS_CU_INTER = 0, XEVE_HIP_BITS_COMP_Y = 1, XEVE_HIP_BITS_COMP_U = 2, XEVE_HIP_BITS_COMP_V = 3, XEVE_HIP_BITS_CU_SKIP = 4,
       XEVE_HIP_BITS_ECO_COEF = 5, /* ctx->fn_eco_coef = xeve_eco_coef (xeve_eco.c:1067-1089) on its own: cbf flags + coefficients */
       XEVE_HIP_BITS_MVP = 6, /* xeve_rdo_bit_cnt_mvp (xeve_mode.c:57-79): mvp_idx + mvd of every used list -- what check_best_mvp prices */
       /* intra CU, Baseline: job.mvp_idx[0] holds the unary index mpm[ipm] of the luma mode (xeve_eco_intra_dir, xeve_eco.c:1104-1121) */
       XEVE_HIP_BITS_CU_INTRA = 7,   /* xeve_rdo_bit_cnt_cu_intra (xeve_mode.c:141-175): skip flag + pred_mode outside I slices, mode, Y / U / V */
       XEVE_HIP_BITS_INTRA_LUMA = 8, /* xeve_rdo_bit_cnt_cu_intra_luma (:81-117): the same with luma alone                                   */
       XEVE_HIP_BITS_INTRA_DIR = 9   /* xeve_rdo_bit_cnt_intra_dir (:136-139): the mode alone                                                */ };
/* XEVE_HIP_BITS_ECO_COEF: job.dir_flag holds these flags.  NO_RESET continues the coder where the entry state stands instead of applying
 * xeve_sbac_bit_reset (needs sbac_out: only the kernel that carries the complete coder state can do it). */
enum { XEVE_HIP_ECO_INTRA = 1, XEVE_HIP_ECO_NO_CBF = 2, XEVE_HIP_ECO_RUN_Y = 4, XEVE_HIP_ECO_RUN_U = 8, XEVE_HIP_ECO_RUN_V = 16, XEVE_HIP_ECO_NO_RESET = 32 };
typedef struct xeve_hip_cu_bits_job {
    int32_t coef_off[3];   /* element offsets of the dense Y / U / V blocks of quantised levels inside `coef` */
    int32_t nnz[3];        /* core->nnz_sub[c][0]: 0 = cbf 0 (the block is not coded whatever it holds)     */
    int32_t sbac;          /* index of the entry state in `sbac_in` (SBAC_LOAD source)                      */
    int16_t mvd[2][2];     /* pi->mvd[pidx]                                                                 */
    int8_t  refi[2];       /* pi->refi[pidx] (< 0 = list unused)                                            */
    uint8_t mvp_idx[2];
    uint8_t mode;          /* XEVE_HIP_BITS_*                                                               */
    uint8_t dir_flag;      /* pidx == PRED_DIR (mode XEVE_HIP_BITS_ECO_COEF: XEVE_HIP_ECO_* flags)                 */
    uint8_t ctx_skip, ctx_pred_mode; /* core->ctx_flags[CNID_SKIP_FLAG], [CNID_PRED_MODE]                   */
} xeve_hip_cu_bits_job;
/* Per job: SBAC_LOAD(sbac_in[job.sbac]) + xeve_sbac_bit_reset + the syntax of job.mode + xeve_get_bit_number -> bits[j];
 * sbac_out (may be NULL) receives the coder state SBAC_STORE would keep, field for field.  coef (coef_elems int16
 * elements), sbac_in, jobs, bits, sbac_out and workspace (>= xeve_hip_cu_bits_workspace(njobs, coef_elems) bytes) are
 * device memory; params is a HOST pointer.  Jobs may share coefficient blocks and entry states.  coef may be NULL (coef_elems 0) when no job
 * codes coefficients (XEVE_HIP_BITS_CU_SKIP / _MVP jobs only): the event pass is then left out. */
size_t xeve_hip_cu_bits_workspace(int njobs, size_t coef_elems);
/* One xeve_eco_coef call in bit-count mode on HOST memory (synchronous): the cbf flags and coefficients of one CU continue the coder from
 * *state where it stands and leave *state as the reference's coder would -- what ctx->fn_eco_coef can be pointed at while sbac->is_bitcount.
 * flags: XEVE_HIP_ECO_INTRA / _NO_CBF / _RUN_Y|U|V; coef_*: the CU's dense coefficient blocks; nnz: core->nnz_sub[c][0]. */
int xeve_hip_eco_coef_host(xeve_hip_sbac *state, const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v, int log2_cuw, int log2_cuh,
                           const int32_t nnz[3], int flags, int chroma_format_idc, int cm_init);
/* The same counts from the count-only kernel, handing on only what later counts depend on: state_out[j].range and .ctx (the other
 * fields are reset, as xeve_sbac_bit_reset would leave them; .code is 0).  For chains of tests such as the per-component cbf
 * tests of pinter_residue_rdo, where the intermediate states are only ever SBAC_LOADed into further bit counts. */
int xeve_hip_cu_bits_jobs_chain(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                                const xeve_hip_cu_bits_params *params, void *workspace, size_t workspace_bytes, uint32_t *bits,
                                xeve_hip_sbac *state_out, void *stream);
int xeve_hip_cu_bits_jobs(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                          const xeve_hip_cu_bits_params *params, void *workspace, size_t workspace_bytes, uint32_t *bits,
                          xeve_hip_sbac *sbac_out, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (5) In-loop deblocking and reference-picture padding: from the        */
/*     reconstruction the residual kernels leave in HBM to the planes the next picture's motion  */
/*     search reads, without leaving the device.  reference: xeve_loop_filter (xeve_enc.c:2355-  */
/*     2415) -> xeve_deblock / xeve_deblock_tree / xeve_deblock_cu_ver / _cu_hor (xeve_df.c),     */
/*     xeve_picbuf_expand (xeve_util.c:190-248).                                                  */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_deblock_params {
    int32_t w, h;                     /* picture size in luma samples */
    int32_t w_scu, h_scu;             /* ctx->w_scu, ctx->h_scu: the maps' dimensions in 4x4 units */
    int32_t log2_max_cuwh;            /* CTU size (unused by the device path: it reads CU sizes from map_cu_mode) */
    int32_t bit_depth_luma, bit_depth_chroma, chroma_format_idc; /* 4:0:0, 4:2:0, 4:4:4 */
    int32_t qp_u_offset, qp_v_offset; /* sh->qp_u_offset / qp_v_offset (pic->pic_qp_*_offset) */
    int32_t qp_chroma[2][100];        /* ctx->qp_chroma_dynamic[c][q] stored at index q + 6 * (bit_depth_chroma - 8) */
} xeve_hip_deblock_params;
/* Both edge directions of one picture (vertical edges first), one slice, quad-tree CUs.  y / u / v point at sample
 * (0, 0) of planes resident in HBM (filtered in place); the maps are the reference's per-4x4-unit arrays, device memory:
 * map_scu (ctx->map_scu: MCU_* bit fields, xeve_def.h:585-640; the COD bits are not used or changed), map_cu_mode
 * (ctx->map_cu_mode: CU log2 width / height in bits 24-31), map_tidx (ctx->map_tidx: the tile of every unit; NULL = one tile): an edge between
 * units of different tiles is not filtered, as in xeve_deblock_cu_hor / _ver (xeve_df.c:296-302,386-391; boundary_filtering is always 0, :528),
 * map_refi [f_scu][2], map_mv [f_scu][2][2] (ctx->map_unrefined_mv, which equals ctx->map_mv in Baseline).  params is a HOST pointer. */
int xeve_hip_deblock(xeve_hip_pel *y, xeve_hip_pel *u, xeve_hip_pel *v, int s_l, int s_c, const uint32_t *map_scu, const uint32_t *map_cu_mode,
                     const uint8_t *map_tidx, const int8_t *map_refi, const int16_t *map_mv, const xeve_hip_deblock_params *params, void *stream);
/* The same on HOST memory, synchronous, whole padded planes staged per call (pad_l / pad_c = how far the buffers extend around
 * the picture): what ctx->fn_loop_filter / ctx->fn_picbuf_expand can be pointed at without the caller owning device memory */
int xeve_hip_deblock_host(xeve_hip_pel *y, xeve_hip_pel *u, xeve_hip_pel *v, int s_l, int s_c, int pad_l, int pad_c, const uint32_t *map_scu,
                          const uint32_t *map_cu_mode, const uint8_t *map_tidx, const int8_t *map_refi, const int16_t *map_mv,
                          const xeve_hip_deblock_params *params);
int xeve_hip_picbuf_expand_host(xeve_hip_pel *y, xeve_hip_pel *u, xeve_hip_pel *v, int s_l, int s_c, int w_l, int h_l, int w_c, int h_c, int exp_l,
                                int exp_c, int chroma_format_idc);
/* xeve_picbuf_expand: replicate the border samples exp_l / exp_c deep around the three planes */
int xeve_hip_picbuf_expand(xeve_hip_pel *y, xeve_hip_pel *u, xeve_hip_pel *v, int s_l, int s_c, int w_l, int h_l, int w_c, int h_c, int exp_l,
                           int exp_c, int chroma_format_idc, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (6) a8: the CU motion-compensation driver.  reference: xeve_mc (src_base/xeve_mc.c:465-610)   */
/*     = xeve_mv_clip (:401-447) + xeve_mc_l / xeve_mc_c per used list (variant from the         */
/*     unclipped vector's fraction, position from the clipped one) + the identical-motion        */
/*     shortcut (:546-551) + xeve_average_16b_no_clip for bi-prediction.                          */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_refpic {
    const xeve_hip_pel *y, *u, *v; /* sample (0, 0) of refp[refi][list].pic's planes (device memory, padded like the reference's) */
    int32_t             poc;       /* refp[refi][list].pic->poc */
    int32_t             pad_;
} xeve_hip_refpic;
typedef struct xeve_hip_cu_mc_job {
    int32_t x, y;      /* CU position in luma samples */
    int16_t mv[2][2];  /* quarter pel */
    int8_t  refi[2];   /* < 0: list unused */
    int8_t  pad_[2];
} xeve_hip_cu_mc_job;
/* refp: HOST array indexed [refi * 2 + list] (entries up to max(num_refp0, num_refp1) - 1); all pictures share s_l / s_c.
 * pred_y [njobs][h*w], pred_u / pred_v [njobs][(h >> h_shift) * (w >> w_shift)] receive what the reference leaves in pred[0];
 * jobs, pred_*, workspace: device memory; coefficient tables: HOST pointers (xeve_tbl_mc_l_coeff / xeve_tbl_mc_c_coeff). */
size_t xeve_hip_mc_cu_workspace(int njobs, int w, int h, int num_refp0, int num_refp1);
/* One xeve_mc call on HOST memory (synchronous; the reference planes the job uses are staged per call): what pi->fn_mc (pinter_mc) can be pointed
 * at.  refp holds HOST plane pointers; the planes extend pad_l / pad_c samples around the picture; pred_* are the caller's pred[0][Y_C / U_C / V_C]. */
int xeve_hip_mc_cu_host(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pad_l, int pad_c, int pic_w, int pic_h,
                        const xeve_hip_cu_mc_job *job, int w, int h, int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc,
                        const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v);
int xeve_hip_mc_cu_jobs(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h,
                        const xeve_hip_cu_mc_job *jobs, int njobs, int w, int h, int bit_depth_luma, int bit_depth_chroma,
                        int chroma_format_idc, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_pel *pred_y,
                        xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (7) The whole of pinter_residue_rdo (src_base/xeve_pinter.c:906-1336) for a batch of inter    */
/*     CU candidates of one size: prediction, residual + SSD, transform + RDOQ with the estimates */
/*     of each candidate's entry coder state, reconstruction + SSD, CABAC bit counts and the      */
/*     coded-block-flag decision (all-zero / as quantised / per component with the coder state    */
/*     handed on / chosen combination), in the reference's double-precision expression order.      */
/*     Presets with rdo_dbk_switch = 0 (fast, medium), no delta QP, tool_iqt 0, CU <= 64x64.        */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_rdo_params {
    int32_t log2_cuw, log2_cuh, pic_w, pic_h;
    int32_t slice_type, num_refp[2], chroma_format_idc, bit_depth, tool_iqt;
    int32_t qp[3];                 /* core->qp_y / qp_u / qp_v */
    int32_t pad_;
    double  lambda[3];             /* core->lambda */
    double  dist_chroma_weight[2]; /* core->dist_chroma_weight */
} xeve_hip_rdo_params;
typedef struct xeve_hip_rdo_job {
    int32_t x, y;
    int16_t mv[2][2], mvd[2][2];   /* pi->mv[pidx], pi->mvd[pidx] */
    int8_t  refi[2];
    uint8_t mvp_idx[2];
    uint8_t dir_flag;              /* pidx == PRED_DIR */
    uint8_t ctx_skip, ctx_pred_mode, pad_;
    int32_t sbac;                  /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] in `states` */
} xeve_hip_rdo_job;
typedef struct xeve_hip_rdo_result {
    double  cost;                  /* the return value of pinter_residue_rdo */
    int32_t nnz[3];                /* core->nnz on exit */
    int32_t pad_;
    int64_t dist[2][3];            /* SSD without residual / as quantised */
} xeve_hip_rdo_result;
/* org (HOST array of three device pointers at sample (0, 0)), refp (HOST, as for xeve_hip_mc_cu_jobs), params, coefficient
 * tables: host memory.  states, jobs, results, coef, best, workspace: device memory.  coef receives pi->coef[pidx]: the Y blocks
 * of all candidates ([njobs][h*w]), then the U blocks, then the V blocks; best[j] = core->s_temp_best (may be NULL: saves one
 * bit-count round). */
size_t xeve_hip_residue_rdo_workspace(int njobs, int nstates, const xeve_hip_rdo_params *params, int s_org_l, int s_org_c);
int xeve_hip_residue_rdo_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                              const xeve_hip_sbac *states, int nstates, const xeve_hip_rdo_params *params, const xeve_hip_rdo_job *jobs, int njobs,
                              const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_rdo_result *results, int16_t *coef,
                              xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream);

/* xeve_analyze_skip (src_base/xeve_pinter.c:1337-1530) for a batch of CUs of one size: every (idx0, idx1) pair of the merge
 * candidate list (pi->mvp / pi->refi_pred as xeve_get_motion left them) that survives the encoder side pruning is predicted, measured
 * (SSD Y + weighted U, V) and priced (skip flag + candidate indices through the CABAC counter from the CU's entry state); the first
 * pair with the strictly smallest cost wins.  rdo_dbk_switch = 0.  In P slices only list 0 is walked (idx1 = 0, refi[1] = -1) and
 * result.mv[1] is pi->mvp[REFP_1][0] as handed in. */
typedef struct xeve_hip_skip_job {
    int32_t x, y;
    int16_t mvp[2][4][2];    /* pi->mvp[list][idx] */
    int8_t  refi_pred[2][4]; /* pi->refi_pred[list][idx] */
    int32_t ncand;           /* pi->skip_merge_cand_num (<= max_cand) */
    int32_t sbac;            /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] in `states` */
    uint8_t ctx_skip, pad_[3];
} xeve_hip_skip_job;
typedef struct xeve_hip_skip_result {
    double  cost;            /* the return value (MAX_COST 1.7e308 when no pair is usable) */
    int64_t best_ssd;        /* pi->best_ssd */
    int32_t idx0, idx1;      /* pi->mvp_idx[PRED_SKIP] */
    int16_t mv[2][2];        /* pi->mv[PRED_SKIP] */
    int8_t  refi[2];         /* pi->refi[PRED_SKIP] */
    int8_t  pad_[6];
} xeve_hip_skip_result;
/* Pointer kinds as for xeve_hip_residue_rdo_jobs.  pred_y [njobs][h*w], pred_u / pred_v [njobs][ch*cw] receive pi->pred[PRED_SKIP][0] of
 * the winner, best[j] (may be NULL) core->s_temp_best; both are left untouched for a CU without a usable pair.  max_cand in 1..4. */
size_t xeve_hip_analyze_skip_workspace(int njobs, const xeve_hip_rdo_params *params, int max_cand);
int xeve_hip_analyze_skip_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                               const xeve_hip_sbac *states, int nstates, const xeve_hip_rdo_params *params, const xeve_hip_skip_job *jobs, int njobs,
                               int max_cand, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_skip_result *results,
                               xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, xeve_hip_sbac *best, void *workspace,
                               size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (8) The whole inter analysis of a CU: xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047) */
/*     = ctx->fn_pinter_analyze_cu, for a batch of CUs of one (square, 8..64) size.  Baseline:      */
/*     skip / merge analysis; unless the skip residual is below the skip_th threshold: temporal   */
/*     direct (B slices), per list the motion search over every reference picture + check_best_mvp */
/*     + pinter_residue_rdo, the iterated bi-prediction search (analyze_bi, B slices) +           */
/*     pinter_residue_rdo; the cheapest mode's coefficients, reconstruction, motion data and      */
/*     coder state.  Every me_algo / me_sub setting; rdo_dbk_switch 0 (presets fast / medium).     */
/* ------------------------------------------------------------------------------------------- */
#define XEVE_HIP_MAX_REFP 8
typedef struct xeve_hip_inter_params {
    xeve_hip_rdo_params  rdo;
    xeve_hip_epzs_params me;   /* lambda_mv, max_search_range, clips, hpel / qpel counts (hpel 0 = ME_LEV_IPEL), me.reserved bit 0 = me_raster on;
                                  bi, extra_bits, refi_bits, range_recentre and the refi of me.reserved are set per search */
    int32_t refi_bits[2][XEVE_HIP_MAX_REFP];      /* xeve_tbl_refi_bits[rdo.num_refp[l]][refi] (xeve_tbl.c:498-517) */
    int32_t range_recentre[2][XEVE_HIP_MAX_REFP]; /* get_range_ipel's POC-distance scaled range of refp[refi][l] (xeve_pinter.c:124-129) */
    int32_t max_cand;                             /* pi->skip_merge_cand_num */
    int32_t poc, col_list_poc0;                   /* ctx->poc.poc_val; refp[0][REFP_1].list_poc[0] (xeve_get_mv_dir, xeve_util.c:634) */
    int32_t pad_;
    double  skip_th;                              /* ctx->param.skip_th */
} xeve_hip_inter_params;
typedef struct xeve_hip_inter_job {
    int32_t x, y;
    int16_t mvp[2][4][2]; /* xeve_get_motion's candidates per list (left, up, up-right, collocated; xeve_util.c:526-573); reference index 0 each */
    int16_t mv_col[2];    /* refp[0][REFP_1].map_mv[bottom-right unit of the CU][0] (xeve_get_mv_dir, xeve_util.c:631-632) */
    int32_t sbac;         /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] in `states` */
    uint8_t ctx_skip, ctx_pred_mode, pad_[2];
} xeve_hip_inter_job;
typedef struct xeve_hip_inter_result {
    double  cost;          /* the return value: cost_inter[best_idx] */
    double  cost_inter[5]; /* PRED_L0, PRED_L1, PRED_BI, PRED_SKIP, PRED_DIR (1.7e308 where not evaluated) */
    int32_t cu_mode;       /* core->cu_mode: MODE_INTER 1 / MODE_SKIP 2 / MODE_DIR 3 */
    int32_t best_idx;      /* PRED_* of the winner */
    int16_t mv[2][2], mvd[2][2]; /* mi->mv, mi->mvd; entries of a list the winner does not use are 0 (stale in the reference) */
    int8_t  refi[2];       /* mi->refi */
    uint8_t mvp_idx[2];    /* mi->mvp_idx (0 for MODE_DIR and unused lists) */
    int32_t nnz[3];        /* core->nnz */
    int32_t pad_[2];
} xeve_hip_inter_result;
/* Pointer kinds as for xeve_hip_residue_rdo_jobs.  coef: the `coef` argument of the reference function, laid out like pi->coef there (Y blocks of
 * all CUs, then U, then V; zero for skipped CUs); rec_y [njobs][w*w], rec_u / rec_v [njobs][cw*ch]: pi->rec[best_idx]; pred_y [njobs][w*w] (may be
 * NULL): mi->pred_y_best; next_best[j]:
 * core->s_next_best[log2_cuw - 2][log2_cuh - 2].  B slices: rdo.num_refp[1] <= rdo.num_refp[0] (analyze_bi walks both lists with num_refp[1]).
 * Asynchronous on `stream` like the rest of the batched API; levels of a picture can be analysed concurrently on separate streams, each call with its
 * own workspace and output buffers (the workspace holds every intermediate of the call).  Capturable into a HIP graph. */
size_t xeve_hip_pinter_analyze_cu_workspace(int njobs, int nstates, const xeve_hip_inter_params *params, int s_org_l, int s_org_c);
int xeve_hip_pinter_analyze_cu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                                    const xeve_hip_sbac *states, int nstates, const xeve_hip_inter_params *params, const xeve_hip_inter_job *jobs,
                                    int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_inter_result *results, int16_t *coef,
                                    xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y, xeve_hip_sbac *next_best,
                                    void *workspace, size_t workspace_bytes, void *stream);
/* The candidates an xeve_hip_inter_job carries, from the maps the encoder keeps per 4x4 unit (device memory): xeve_get_avail_inter
 * (xeve_util.c:652-714; the left / up / up-right bits) + xeve_get_motion (xeve_util.c:526-573) per list + the collocated vector xeve_get_mv_dir
 * reads (xeve_util.c:631-632, the CU's bottom-right unit).  map_scu: ctx->map_scu; map_tidx: ctx->map_tidx (NULL = one tile); map_mv: ctx->map_mv
 * ([unit][list][x, y]); col_mv0 / col_mv1: refp[0][REFP_0 / REFP_1].map_mv.  jobs[j].x / y are read, jobs[j].mvp / mv_col are written (list 1 and
 * mv_col stay 0 in P slices); the other fields are the caller's. */
int xeve_hip_inter_candidates(const uint32_t *map_scu, const uint8_t *map_tidx, const int16_t *map_mv, const int16_t *col_mv0, const int16_t *col_mv1, int w_scu,
                              int h_scu, int log2_cuw, int log2_cuh, int slice_type, xeve_hip_inter_job *jobs, int njobs, void *stream);
/* Resident pictures (serving layer, first slice).  The reference keeps its pictures in host memory and calls the path once per CU; a caller that
 * announces every new picture with xeve_hip_picture_begin() -- the reference's hook is ctx->fn_mode_analyze_frame, called once per picture before the
 * CTU loop (src_base/xeve_enc.c:275, xeve_mode.c:2441) -- gets each host plane uploaded ONCE per picture (the first time a host-memory entry point sees
 * it) and served from HBM for all later calls of that picture.  Contract: no host-memory call in flight during xeve_hip_picture_begin(); planes do not
 * change between two of them.  Never called: every host-memory call stages its planes itself (the round-1 behaviour).  _stats: pictures announced, planes
 * uploaded (+ bytes) and cache hits since init. */
int xeve_hip_picture_begin(void);
/* Leaves resident mode (every later host-memory call stages its planes itself) until the next xeve_hip_picture_begin(): for a caller that stops announcing pictures. */
int xeve_hip_picture_end(void);
int xeve_hip_resident_stats(uint64_t *pictures, uint64_t *uploads, uint64_t *upload_bytes, uint64_t *hits);

/* One xeve_pinter_analyze_cu call on HOST memory (synchronous; without resident pictures the original and every reference picture of both lists are staged per call): what
 * ctx->fn_pinter_analyze_cu can be pointed at.  org / refp: HOST pointers to sample (0, 0); the reference planes extend pad_l / pad_c samples around
 * the picture; *state: core->s_curr_best[log2_cuw - 2][log2_cuh - 2] (job->sbac is ignored); coef_* / rec_*: the CU's dense blocks. */
int xeve_hip_pinter_analyze_cu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, int pad_l, int pad_c,
                                    const xeve_hip_sbac *state, const xeve_hip_inter_params *params, const xeve_hip_inter_job *job, const int16_t (*coef_l)[8],
                                    const int16_t (*coef_c)[4], xeve_hip_inter_result *result, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                                    xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y, xeve_hip_sbac *next_best);

/* One pinter_me_epzs call on HOST memory (synchronous; both luma planes staged per call): what pi->fn_me can be pointed at.
 * org0 / ref0 = sample (0, 0) of the original / reference luma plane; the reference plane has `pad` samples around the picture. */
int xeve_hip_me_epzs_host(const xeve_hip_pel *org0, int s_org, const xeve_hip_pel *org_bi, const xeve_hip_pel *ref0, int s_ref, int pad, int pic_h,
                          const xeve_hip_epzs_job *job, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                          const xeve_hip_epzs_params *params, xeve_hip_me_result *result);

/* ------------------------------------------------------------------------------------------- */
/* The intra analysis of a batch of CUs of one size: pintra_analyze_cu (src_base/xeve_pintra.c:544-698)  */
/* = ctx->fn_pintra_analyze_cu.  Baseline profile, rdo_dbk_switch 0, no delta QP, square CUs 4..64.  */
/* Neighbour samples from the picture being reconstructed + the 4x4-unit maps (xeve_get_nbr), the    */
/* five predictors, the SATD + mode-bits candidate list cut against the best inter prediction's    */
/* SATD (make_ipred_list), the luma RDO of the list, the chroma RDO of its winner                  */
/* (pintra_residue_rdo), the CU's cost from the whole intra syntax.                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_intra_params {
    int32_t log2_cuw, log2_cuh, w_scu, h_scu;   /* picture size in 4x4 units (ctx->w_scu, ctx->h_scu) */
    int32_t slice_type, chroma_format_idc, bit_depth, tool_iqt;
    int32_t constrained_intra_pred, qp[3];      /* pps.constrained_intra_pred_flag; core->qp_y / qp_u / qp_v */
    double  lambda[3];                          /* core->lambda */
    double  sqrt_lambda0;                       /* core->sqrt_lambda[0] */
    double  dist_chroma_weight[2];
} xeve_hip_intra_params;
typedef struct xeve_hip_intra_job {
    int32_t  x, y;
    uint32_t inter_satd;   /* core->inter_satd: SATD of the best inter prediction, 0xFFFFFFFF without one (mode_check_intra, xeve_mode.c:1250-1262) */
    int32_t  sbac;         /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] in `states` */
    int32_t  pic;          /* picture of a multi-picture batch (0 with pic_elems == NULL) */
    uint8_t  ctx_skip, ctx_pred_mode, pad_[2];
} xeve_hip_intra_job;
typedef struct xeve_hip_intra_result {
    double  cost;          /* the return value */
    int32_t dist_cu;       /* core->dist_cu */
    int32_t nnz[3];        /* core->nnz */
    int32_t pred_cnt;      /* candidates that went through the luma RDO */
    int8_t  ipm[2];        /* core->ipm */
    int8_t  pad_[2];
} xeve_hip_intra_result;
/* org / mod: HOST arrays of three device pointers at sample (0, 0): the original planes and the planes of the picture being
 * reconstructed (pi->m = PIC_MODE(ctx)); map_scu / map_ipm / map_tidx: ctx->map_scu, ctx->map_ipm, ctx->map_tidx (device).
 * pic_elems (HOST, or NULL for one picture): element distance between consecutive pictures of a batch that spans several --
 * {org luma, org chroma, mod luma, mod chroma, maps}; job.pic selects the picture.  params: host.  states, jobs, results, coef, rec,
 * best, workspace: device.  coef receives the `coef` argument: the Y blocks of all CUs ([njobs][h*w]), then U, then V; rec
 * receives pi->rec in the same layout (dense blocks); best[j] = core->s_temp_best (may be NULL). */
size_t xeve_hip_pintra_analyze_cu_workspace(int njobs, int nstates, const xeve_hip_intra_params *params);
int xeve_hip_pintra_analyze_cu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                    const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, const int64_t *pic_elems,
                                    const xeve_hip_sbac *states, int nstates, const xeve_hip_intra_params *params, const xeve_hip_intra_job *jobs, int njobs,
                                    xeve_hip_intra_result *results, int16_t *coef, xeve_hip_pel *rec, xeve_hip_sbac *best, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* Host-memory form of ONE call of ctx->fn_pintra_analyze_cu: every pointer is HOST memory -- org / mod = sample (0, 0) of the original picture's planes and of
 * the picture being reconstructed (pi->o / pi->m with their strides), the maps = ctx->map_scu / map_ipm / map_tidx, state = core->s_curr_best[..][..] (job->sbac
 * and job->pic are ignored).  Moves the CU's block of the original, the line above and the column left of the CU (cuw + cuh samples each, clipped to the
 * picture) and the map entries of their 4x4 units; returns what the reference's function leaves behind (rec_* = pi->rec, dense; best = core->s_temp_best). */
int xeve_hip_pintra_analyze_cu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                    const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, const xeve_hip_sbac *state,
                                    const xeve_hip_intra_params *params, const xeve_hip_intra_job *job, xeve_hip_intra_result *result, int16_t *coef_y, int16_t *coef_u,
                                    int16_t *coef_v, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_sbac *best);

/* ------------------------------------------------------------------------------------------- */
/* The mode decision of a batch of CTUs of I pictures: mode_analyze_lcu -> mode_coding_tree          */
/* (src_base/xeve_mode.c:2007-2375, 2518-2610) = ctx->fn_mode_analyze_lcu for an I slice, on the device.*/
/* Baseline quad-tree, no delta QP, rdo_dbk_switch 0.  Every chain (one CTU of one picture) walks its */
/* tree in the reference's order -- the CU as a whole (split_cu_flag = 0 + the intra analysis above), */
/* the early-termination rule of I pictures, the four quadrants (split_cu_flag = 1), the cheaper one   */
/* kept -- updating the picture being reconstructed and the 4x4-unit maps as it goes (clear_map_scu     */
/* :1129, copy_to_cu_data :868, update_map_scu :1036, mode_cpy_rec_to_ref :797), because the next CU's   */
/* neighbours are read from them.  Chains of one call must not overlap (different pictures, or CTUs   */
/* that are not neighbours); they advance in lockstep, one batched intra analysis per tree node.       */
/* ------------------------------------------------------------------------------------------- */
#define XEVE_HIP_CU_DEPTHS 10 /* cud = 0, 2, .. 8 for 64 .. 4 (a quad split counts as two levels, xeve_util.c:1407-1410): rows of XEVE_CU_DATA.split_mode[][SQUARE][] */
typedef struct xeve_hip_tree_params {
    xeve_hip_intra_params ip;   /* what every CU's intra analysis gets (log2_cuw / log2_cuh are set per node) */
    int32_t pic_w, pic_h;       /* ctx->w, ctx->h */
    int32_t log2_ctu;           /* ctx->log2_max_cuwh (3 .. 6) */
    int32_t max_cu, min_cu;     /* ctx->param.max_cu_intra, min_cu_intra in an I slice, max_cu_inter, min_cu_inter otherwise (samples; xeve_mode.c:2047-2054).  min_cu 4 in
                                   an inter slice (preset placebo: 4x4 inter CUs): fused walk only */
    int32_t min_cuwh;           /* ctx->min_cuwh */
    int32_t slice_qp, slice_num;/* ctx->tile[].qp (the QP field of map_scu), ctx->slice_num */
    int32_t rdo_dbk;            /* ctx->param.rdo_dbk_switch (presets slow, placebo): every candidate's distortion includes what the loop filter will do to its left / top
                                   boundary (calc_delta_dist_filter_boundary, xeve_mode.c:1534-2005).  Fused walk only: the composed walk refuses 1 */
} xeve_hip_tree_params;
typedef struct xeve_hip_ctu_job {
    int32_t x, y;   /* core->x_pel, core->y_pel */
    int32_t sbac;   /* index of core->s_curr_best[log2_max_cuwh - 2][log2_max_cuwh - 2] in `states` */
    int32_t pic;    /* picture of a multi-picture batch (0 with pic_elems == NULL) */
} xeve_hip_ctu_job;
typedef struct xeve_hip_ctu_data { /* the fields of XEVE_CU_DATA (xeve_type.h:573-617) an I-slice CTU carries; 4x4 units in raster order, pitch = the CTU's width in units */
    int8_t   split_mode[XEVE_HIP_CU_DEPTHS][256]; /* [depth][unit], shape SQUARE */
    uint8_t  pred_mode[256];
    int8_t   ipm[2][256], depth[256];
    int32_t  nnz[3][256];
    uint32_t map_scu[256], map_cu_mode[256];
    int16_t  coef[3][64 * 64];
    xeve_hip_pel reco[3][64 * 64];
    int16_t  mv[256][2][2], mvd[256][2][2]; /* P / B slices: [unit][list][x, y]; zero for an intra unit and for a list the CU does not use */
    int8_t   refi[256][2];                  /* (-1, -1) for an intra unit */
    uint8_t  mvp_idx[256][2];
} xeve_hip_ctu_data;
/* The inter side of the walk (P / B slices): mode_coding_unit (xeve_mode.c:1310-1350) = mode_check_inter (the whole inter analysis above) then, when
 * the inter winner has a residual, mode_check_intra with the candidate list cut against the SATD of the inter winner's luma prediction; a skipped CU
 * at depth >= ecu_depth ends the split (:2162-2172).  Baseline (tool_admvp 0), square CUs 8 .. 64 (min_cu_inter >= 8). */
typedef struct xeve_hip_tree_inter {
    const xeve_hip_refpic *refp;        /* HOST table [refi * 2 + list] of device planes, as for xeve_hip_pinter_analyze_cu_jobs */
    int32_t                s_ref_l, s_ref_c;
    xeve_hip_inter_params  ipar;        /* rdo.log2_cuw / log2_cuh are set per node */
    int16_t               *map_mv;      /* device, ctx->map_mv [unit][list][x, y]: read for the candidates, updated with every decided CU */
    int8_t                *map_refi;    /* device, ctx->map_refi [unit][list]: updated */
    const int16_t         *col_mv0, *col_mv1; /* device, refp[0][REFP_0 / REFP_1].map_mv */
    const int16_t        (*coef_l)[8];  /* HOST: the interpolation filters (pi->mc_l_coeff / mc_c_coeff) */
    const int16_t        (*coef_c)[4];
    int32_t                ecu_depth;   /* ENC_ECU_DEPTH_B 4, minus 2 on odd POCs (ENC_ECU_ADAPTIVE) */
    int32_t                pad_;
} xeve_hip_tree_inter;
/* org / mod: HOST arrays of three device pointers at sample (0, 0); mod (PIC_MODE(ctx)), map_scu, map_ipm, map_cu_mode are read AND updated; on return
 * they hold the CTUs' decisions as after update_to_ctx_map + update_map_scu (the caller's reset of the coded flags, xeve_mode.c:2591-2607, is not applied).
 * pic_elems as in xeve_hip_pintra_analyze_cu_jobs (map_cu_mode strides like map_scu).  states, jobs, out, next_best (core->s_next_best[..][..]), cost
 * (the tree's cost), workspace: device. */
size_t xeve_hip_mode_analyze_ctu_intra_workspace(int nchains, const xeve_hip_tree_params *params);
/* Every slice type: params->ip.slice_type 0 B / 1 P with `inter` (max_cu / min_cu = ctx->param.max_cu_inter / min_cu_inter), 2 I with inter == NULL (the
 * function below).  P / B: the chains of one call belong to ONE picture (jobs[].pic 0, pic_elems NULL) -- e.g. the CTU rows of a wavefront.
 * _workspace's nchains is the WIDTH OF THE BATCH: max(nchains, nstates) over the calls that will use the workspace (the library picks the walk -- and with it the
 * workspace layout -- by that width, xeve_hip_walk_fused; a run whose workspace is too small for the layout it picks returns XEVE_HIP_ERR_ARG). */
size_t xeve_hip_mode_analyze_ctu_workspace(int nchains, const xeve_hip_tree_params *params, const xeve_hip_tree_inter *inter, int s_org_l, int s_org_c);
int xeve_hip_mode_analyze_ctu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                   uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems,
                                   const xeve_hip_sbac *states, int nstates, const xeve_hip_tree_params *params, const xeve_hip_tree_inter *inter,
                                   const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, void *workspace,
                                   size_t workspace_bytes, void *stream);
/* The walk runs as ONE kernel per call (xeve_amd/csrc/walk.h: a workgroup per team of chains executes the whole quad-tree schedule; XEVE_HIP_WALK=0 selects the composed
 * walk of ~10 000 launches per step instead).  Its in-kernel stage profile: _enable(1) makes thread 0 of team 0 add the shader cycles between two stage marks to the
 * stage's class; _prof copies {cycles[n], marks[n]} of the classes since the last call into out (cap >= 2 n entries) and returns n (0: the profile is off). */
int xeve_hip_walk_prof_enable(int on);
int xeve_hip_walk_prof(unsigned long long *out, int cap);
/* 1: a call in a batch of nchains chains (max(nchains, nstates) of the call) runs the fused walk, 0: the composed walk (XEVE_HIP_WALK=1 / 0 pins one; unset: fused up to
 * XEVE_HIP_WALK_AUTO_MAX chains -- 0 since round 6: with its side stream the composed walk finishes a step sooner at every width; presets slow and placebo run on the
 * fused kernel whatever the width). */
int xeve_hip_walk_fused(int nchains);
/* Moves that choice at run time: mode -1 by the width (the default), 0 the composed walk, 1 the fused kernel; returns the mode before the call (any other value only
 * reads it).  Process-wide; workspaces are sized by the choice in force when xeve_hip_mode_analyze_ctu_workspace / xeve_hip_enc_create is called, so select BEFORE
 * creating an encoder or sizing a workspace and keep it until that object is gone. */
int xeve_hip_walk_select(int mode);
/* Chains a team of the fused kernel carries: 0 (the default) = as few as keep every chain of the running encoders resident (1 up to ~1000 chains), 1..8 pins it
 * (XEVE_HIP_WALK_C); returns the value before the call (a value outside 0..8 only reads it).  Results do not depend on it. */
int xeve_hip_walk_team(int chains_per_team);
/* The composed walk's SIDE STREAM (round 6): the analyses of every node that has children (the unsplit alternative of mode_coding_tree, xeve_mode.c:2073-2146) run on a
 * second stream of the library's while the caller's stream walks on into the children (:2189-2262) -- neither needs anything of the other until the two costs are compared
 * (:2306-2329) --, joined with events in front of that comparison: two launch chains side by side.  1 on (the default; XEVE_HIP_TREE_SIDE=0 starts with it off), 0 the
 * one-stream walk, 2 a side stream per node size (a measurement setting: slower); returns the value before the call (any other value only reads it).  Results do not
 * depend on it; the workspace query covers all three. */
int xeve_hip_walk_side(int on);
int xeve_hip_mode_analyze_ctu_intra_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                         uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems,
                                         const xeve_hip_sbac *states, int nstates, const xeve_hip_tree_params *params, const xeve_hip_ctu_job *jobs, int nchains,
                                         xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, void *workspace, size_t workspace_bytes, void *stream);
/* Host-memory form of ONE call of ctx->fn_mode_analyze_lcu in an I slice (one host<->device exchange per CTU): every pointer is HOST memory -- org / mod =
 * sample (0, 0) of the original picture's planes and of PIC_MODE(ctx) with their strides, the maps = ctx->map_scu / map_ipm / map_tidx / map_cu_mode, entry =
 * core->s_curr_best[log2_max_cuwh - 2][log2_max_cuwh - 2], (x0, y0) = core->x_pel, y_pel.  Moves the CTU, one unit to its left and above and the CTU's width
 * to its right (clipped to the picture); writes the CTU's reconstruction and map entries back, out = what the walk leaves in core->cu_data_best[..][..]. */
int xeve_hip_mode_analyze_ctu_intra_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                         uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xeve_hip_sbac *entry,
                                         const xeve_hip_tree_params *params, int x0, int y0, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost);
/* The same for every slice type.  I slices (inter == NULL): the function above.  P / B slices: `inter` with HOST pointers throughout -- refp[].y/u/v = sample
 * (0, 0) of the reference pictures' host planes, which extend pad_l / pad_c samples around the picture; map_mv / map_refi = ctx->map_mv / map_refi (read and
 * updated); col_mv0 / col_mv1 = refp[0][REFP_0 / REFP_1].map_mv -- and resident pictures (xeve_hip_picture_begin once per picture): the original, the reference
 * pictures, the collocated maps and the tile map are uploaded once per picture, the CTU's neighbourhood of PIC_MODE and of the maps per call. */
int xeve_hip_mode_analyze_ctu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                   uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xeve_hip_sbac *entry,
                                   const xeve_hip_tree_params *params, const xeve_hip_tree_inter *inter, int pad_l, int pad_c, int x0, int y0,
                                   xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost);

/* ------------------------------------------------------------------------------------------- */
/* The bitstream writer's side of a batch of decided CTUs: xeve_eco_tree (src_base/xeve_enc.c:35-100) */
/* -> xeve_eco_split_mode, xeve_eco_unit (xeve_eco.c:1377-1640) for every CU of each chain's CTU, on */
/* the WRITER's coder: states[jobs[c].sbac] is read, advanced and stored back -- it then is the state */
/* the chain's next CTU starts its mode decision from (xeve_enc.c:139).  The syntax is the writer's,  */
/* not the rate estimate's (P slices: no direct_mode_flag, no inter_pred_idc).  The coded flags of    */
/* the CTU's units are reset first (xeve_mode.c:2591-2607), then every CU stores what xeve_eco_unit   */
/* stores (coded / skip / luma cbf flags, CU size).  Baseline, no delta QP.                          */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_eco_params {
    int32_t chroma_format_idc, slice_type, log2_ctu, pic_w, pic_h, w_scu, h_scu;
    int32_t num_refp[2];  /* ctx->rpm.num_refp (P / B slices) */
    int32_t pad_;
} xeve_hip_eco_params;
/* ctus[c]: the CTU xeve_hip_mode_analyze_ctu_jobs decided for chain c (jobs as there); bytes: [nchains][bytes_cap], nbytes[c] = bytes the coder emitted for chain
 * c (the first bytes_cap of them stored; the pending byte and the code register stay in the state); map_pic_elems: element distance between the pictures' maps
 * (0: one picture).  Everything but params is device memory. */
int xeve_hip_eco_ctu_jobs(const xeve_hip_ctu_data *ctus, xeve_hip_sbac *states, int nstates, const xeve_hip_eco_params *params, uint32_t *map_scu, const int8_t *map_ipm,
                          const uint8_t *map_tidx, uint32_t *map_cu_mode, int64_t map_pic_elems, const xeve_hip_ctu_job *jobs, int nchains, uint8_t *bytes, int bytes_cap,
                          int32_t *nbytes, void *stream);
/* The end of a tile on each chain's writer state (states[jobs[c].sbac], in place): xeve_eco_tile_end_flag(bs, 1) and xeve_sbac_finish (xeve_eco.c:577-595, 622-672).
 * bytes: [nchains][bytes_cap], nbytes[c] = what comes out -- appended to the bytes of the chain's CTUs it completes the slice data of the tile. */
int xeve_hip_eco_tile_end_jobs(xeve_hip_sbac *states, int nstates, const xeve_hip_ctu_job *jobs, int nchains, uint8_t *bytes, int bytes_cap, int32_t *nbytes, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* The closed-GOP batch encoder: the reference's encoder API (inc/xeve.h: xeve_create :441,      */
/* xeve_push :443, xeve_encode :444, xeve_delete :442; the frame loop xeve_enc / xeve_pic,        */
/* src_base/xeve_enc.c:226-640) for NGOPS independent encoder runs at once.  Every run codes one  */
/* closed GOP of `frames` pictures exactly as `xeveb_app --seek g*frames --frames frames` does    */
/* (the concatenation of the runs is the bitstream of the whole sequence); the    */
/* runs advance in lockstep through the device entry points above (CTU mode decision -> writer    */
/* -> tile end -> loop filter -> padding), every picture resident in HBM.  Baseline profile,      */
/* constant QP, all four presets (slow and placebo on the fused CTU walk: rdo_dbk_switch and     */
/* 4x4 inter CUs live there), 4:2:0, 8- or 10-bit input coded at 10 bits -- the application's      */
/* defaults.  `threads` is the reference's -m: the CTU-row chains of a picture (the bitstream     */
/* depends on it, as the reference's does).                                                       */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_enc_config {
    int32_t w, h;             /* -w / -h: multiples of 8 */
    int32_t fps_num, fps_den; /* -z (only the SEI text carries it) */
    int32_t qp;               /* -q */
    int32_t keyint;           /* -I */
    int32_t bframes;          /* -b: 0, 1, 3, 7, 15 */
    int32_t closed_gop;       /* --closed-gop */
    int32_t preset;           /* 0 fast, 1 medium, 2 slow, 3 placebo (xeve_param_apply_ppt_baseline, xeve_enc.c:2431-2506) */
    int32_t threads;          /* -m: 1 .. 8 */
    int32_t inter_slice_type; /* --inter-slice-type: 0 B, 1 P */
    int32_t ref;              /* --ref (0: the preset's) */
    int32_t reserved[4];      /* [0] bit 0: always run the second writer pass (tests); bit 1: the application's --info 0 (no SEI with the option list); bits 8-15: --level-idc (0: 40).  [1]: the application's -d, the input's bit depth -- 0 / 8: one byte per sample;
                               * 10: 16-bit little-endian samples (frames pushed are twice as long).  Either way the codec works at 10 bits (--codec-bit-depth).  [2], [3]: --qp-cb-offset / --qp-cr-offset, -12 .. 12
                               * (pinned against the reference LIBRARY: the application lists the options and cannot parse them; 0 with presets slow / placebo). */
} xeve_hip_enc_config;
typedef struct xeve_hip_enc xeve_hip_enc;
/* A batch of `ngops` runs of `frames` pictures each.  NULL + xeve_hip_last_error() when the configuration is outside the supported set or HBM does not hold the batch. */
xeve_hip_enc *xeve_hip_enc_create(const xeve_hip_enc_config *cfg, int ngops, int frames);
void          xeve_hip_enc_delete(xeve_hip_enc *e);
/* Frame `frame` of run `gop`: planar 4:2:0 (w*h luma samples, then U, then V; one byte each, or two with an input depth of 10), host memory (on_device 0) or device
 * memory (1).  The frame is copied into HBM
 * before the call returns (as xeve_push copies its image); the copy runs on the encoder's own stream, so device memory must be COMPLETE when the call is made --
 * work still queued on another stream that produces it is not waited for. */
int xeve_hip_enc_push(xeve_hip_enc *e, int gop, int frame, const uint8_t *yuv, int on_device);
/* Codes every run (synchronous).  A run CONSUMES its frames -- the picture stores of later pictures reuse the memory of frames already coded -- so every frame
 * must be pushed again before the next run of the same object (xeve_hip_enc_begin refuses otherwise). */
int xeve_hip_enc_encode(xeve_hip_enc *e);
/* The same in slices: xeve_hip_enc_begin, then xeve_hip_enc_advance until *remaining is 0.  The unit is the LOCKSTEP STEP -- one CTU of every row chain of every
 * run decided and written (xeve_ctu_mt_core's loop body, xeve_enc.c:128-170) -- a picture's set-up rides on its first step, its end (loop filter, slice data, NAL
 * units) on its last.  advance returns when max_steps more steps are ISSUED; xeve_hip_enc_sync waits for the device (what a timer needs around a slice). */
int xeve_hip_enc_begin(xeve_hip_enc *e);
int xeve_hip_enc_advance(xeve_hip_enc *e, int64_t max_steps, int64_t *remaining);
int xeve_hip_enc_sync(xeve_hip_enc *e);
/* A picture's access unit is appended to its run's bitstream when the NEXT picture ends (its second writer pass runs beside that picture's steps).  _flush appends the
 * one still outstanding now (waits for it): afterwards xeve_hip_enc_bitstream holds every picture whose steps have all been issued -- for a caller that stops part way
 * (a bounded measurement, a check against the reference's bitstream after k pictures).  The run may be advanced further afterwards. */
int xeve_hip_enc_flush(xeve_hip_enc *e);
/* Run `gop`'s bitstream (what the application would have written to its output file); valid until the next encode / delete. */
int xeve_hip_enc_bitstream(xeve_hip_enc *e, int gop, const uint8_t **data, size_t *bytes);
/* Lockstep statistics of the last encode: CTU steps issued, seconds inside the step calls / the picture-end calls (loop filter, second writer pass, padding). */
int xeve_hip_enc_stats(xeve_hip_enc *e, int64_t *ctu_steps, double *step_seconds, double *picture_end_seconds);
/* How large may a batch be, and what does it cost?  *max_gops: the runs one batch can hold at this picture size (its stacked originals are addressed in pairs of samples with 32 bits:
 * 2^33 samples, 896 pictures of 3840x2160); *device_bytes: the HBM a batch of `ngops` x `frames` takes between create and delete (picture stores, maps, both CTU stores, the
 * walk's workspace).  No device call: usable before xeve_hip_init.  A job larger than one batch is several xeve_hip_enc objects, a host thread each -- the device runs
 * their launch chains side by side (xeve_amd/encode.py encode_gops does the split). */
int xeve_hip_enc_footprint(const xeve_hip_enc_config *cfg, int ngops, int frames, uint64_t *device_bytes, int32_t *max_gops);

/* ------------------------------------------------------------------------------------------- */
/* Main profile: the adaptive loop filter's sample kernels ("ALF           */
/* classification / filter").  reference: src_main/xevem_alf.c -- alf_copy_and_extend            */
/* (:91-168), alf_derive_classification / _blk (:463-654), alf_filter_blk_7 / _5 (:656-882;       */
/* the ADAPTIVE_LOOP_FILTER object's derive_classification_blk / filter_7x7_blk /                 */
/* filter_5x5_blk pointers, :50-53, xevem_alf.h:264-266), xeve_alf_get_blk_stats +                */
/* xeve_alf_clac_covariance (:3836-3952).  Planes, classifier and job lists are DEVICE            */
/* memory; the filter derivation between statistics and filtering (Cholesky solves, rate          */
/* estimates: xeve_alf_encode) stays the caller's.                                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_alf_area { int32_t x, y, w, h; } xeve_hip_alf_area; /* AREA (xevem_alf.h:65-71): multiples of 4 */
typedef struct xeve_hip_alf_filter_job {
    int32_t x, y, w, h;       /* the area: its position in the classifier plane (7-tap filter; ignored by the 5-tap one) and its size, multiples of 4 */
    int64_t dst_off, src_off; /* elements from `dst` / `src` to the area's first sample (the reference passes pre-offset pointers: rec_dst, rec_src) */
} xeve_hip_alf_filter_job;
typedef struct xeve_hip_alf_clip_range { int32_t min, max, bd, n; } xeve_hip_alf_clip_range; /* CLIP_RANGE (xevem_alf.h:89-95) */
/* alf_copy_and_extend / alf_copy_and_extend_tile: the w x h samples at rec -> tmp (both point at sample (0, 0) of the area), then m samples of edge replication all round */
int xeve_hip_alf_copy_and_extend(xeve_hip_pel *tmp, int s_tmp, const xeve_hip_pel *rec, int s_rec, int w, int h, int m, void *stream);
/* alf_derive_classification over one area (HOST pointer; picture coordinates): classifier[(y + i) * s_cls + x + j] = (class << 2) | transposition of the sample's 4x4
 * block.  src_luma at sample (0, 0) of the picture the area counts in; reads 3 samples around the area (the class of a block is a function of the 10 x 10 samples around
 * it, so any partition into areas gives the same plane).  classifier 4-byte aligned, s_cls a multiple of 4. */
int xeve_hip_alf_classify(uint8_t *classifier, int s_cls, const xeve_hip_pel *src_luma, int s_src, const xeve_hip_alf_area *area, int bit_depth, void *stream);
/* alf_filter_blk_7 (taps 7: per 4x4 block the 13 coefficients of its class -- filter_set[25][13], HOST memory -- in its transposition) / alf_filter_blk_5 (taps 5:
 * filter_set[7], classifier unused) of every job; src carries 3 samples of margin around every area (the CTU window xeve_alf_recon cuts, :2133-2170, or the extended
 * picture); dst and src must not overlap.  out = clip(clip_min, clip_max, (sum + 256) >> 9). */
int xeve_hip_alf_filter_jobs(int taps, xeve_hip_pel *dst, int s_dst, const xeve_hip_pel *src, int s_src, const uint8_t *classifier, int s_cls,
                             const xeve_hip_alf_filter_job *jobs, int njobs, const int16_t *filter_set, int clip_min, int clip_max, void *stream);
/* xeve_alf_get_blk_stats of every job (an area in picture coordinates; org / rec / classifier at sample (0, 0); rec with 3 samples of margin): per class the
 * auto-correlation of the local sums, E [njobs][nclasses][13][13] (full, symmetric; zero beyond the shape's coefficients: taps 7 -> 13, 5 -> 7), their
 * cross-correlation with org - rec, y [njobs][nclasses][13], and the energy pix_acc [njobs][nclasses].  nclasses = 25 with a classifier, 1 without (chroma: class 0,
 * no transposition).  WRITTEN, not accumulated (the reference adds into records it has just reset, :3754-3762): exact integers, so sums over jobs are exact too. */
int xeve_hip_alf_blk_stats_jobs(int taps, const uint8_t *classifier, int s_cls, const xeve_hip_pel *org, int s_org, const xeve_hip_pel *rec, int s_rec,
                                const xeve_hip_alf_area *jobs, int njobs, double *E, double *y, double *pix_acc, void *stream);
/* HOST-memory forms with the reference's own signatures: what alf->derive_classification_blk / filter_7x7_blk / filter_5x5_blk (set in alf_init, :50-53) can be
 * pointed at, and what xeve_alf_derive_stats_filtering can call instead of xeve_alf_get_blk_stats (:3810; the filter shape replaced by its length).  Synchronous, the area and its margin staged per call; like the table functions of section (1) they cannot report failure and abort with a message. */
typedef struct xeve_hip_alf_covariance { int32_t num_coef; double *y; double **E; double pix_acc; } xeve_hip_alf_covariance; /* ALF_COVARIANCE (xevem_alf.h:291-297) */
/* xeve_alf_get_blk_stats (:3836-3888) on host memory: alf_cov[25] (classifier given) or alf_cov[1]; taps = shape->filterLength (5 | 7).  As the reference: the upper
 * triangle of E, y and pix_acc are ADDED to, then every class's lower triangle is set from its upper one. */
void xeve_hip_alf_get_blk_stats_host(int taps, xeve_hip_alf_covariance *alf_cov, uint8_t **classifier, const xeve_hip_pel *org0, int org_stride, const xeve_hip_pel *rec0,
                                     int rec_stride, int x, int y, int width, int height);
void xeve_hip_alf_derive_classification_blk_host(uint8_t **classifier, const xeve_hip_pel *src_luma, int src_stride, const xeve_hip_alf_area *blk, int shift, int bit_depth);
void xeve_hip_alf_filter_blk_7_host(uint8_t **classifier, xeve_hip_pel *rec_dst, int dst_stride, const xeve_hip_pel *rec_src, int src_stride, const xeve_hip_alf_area *blk,
                                    uint8_t comp_id, short *filter_set, const xeve_hip_alf_clip_range *clip_range);
void xeve_hip_alf_filter_blk_5_host(uint8_t **classifier, xeve_hip_pel *rec_dst, int dst_stride, const xeve_hip_pel *rec_src, int src_stride, const xeve_hip_alf_area *blk,
                                    uint8_t comp_id, short *filter_set, const xeve_hip_alf_clip_range *clip_range);

/* ------------------------------------------------------------------------------------------- */
/* Main profile: affine motion compensation of a batch of CUs of one size (*/
/* "affine MC").  reference: xeve_affine_mc (src_main/xevem_mc.c:2236-2339) =                      */
/* derive_affine_subblock_size_bi (xevem_util.c:1203-1272), per list in use xeve_affine_mc_lc      */
/* (:1671-1915: the Main 8- / 4-tap filters, or -- sub-blocks below 8 -- the enhanced              */
/* interpolation filter, xeve_eif_mc :2123-2234), the average of two lists.  4:2:0.                */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_affine_job {
    int32_t x, y;         /* CU position in luma samples (multiples of 4) */
    int16_t mv[2][3][2];  /* mv[list][VER_NUM][MV_D]: the control-point vectors (top-left, top-right, bottom-left), quarter pel */
    int8_t  refi[2];      /* < 0: list unused; else below the list's num_refp */
    int8_t  vertex_num;   /* 2: four-parameter model (the third vector is not read) | 3 */
    int8_t  pad_;
} xeve_hip_affine_job;
/* refp: HOST array indexed [refi * 2 + list] of DEVICE planes (sample (0, 0); padded like the reference's: the vectors are clipped to 128 samples around the picture,
 * the filters reach 4 further); jobs, pred_*: device memory.  pred_y [njobs][h * w], pred_u / pred_v [njobs][(h / 2) * (w / 2)] receive what the reference leaves in
 * pred[0][Y_C / U_C / V_C].  w, h: powers of two, 8 .. 128. */
int xeve_hip_affine_mc_jobs(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h, const xeve_hip_affine_job *jobs, int njobs,
                            int w, int h, int bit_depth, xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, void *stream);
/* Host-memory form of ONE xeve_affine_mc call with the reference's arguments (src_main/xevem_mc.h:106-120): refi / mv as the encoder holds them, refp = HOST table
 * [refi * 2 + list] of HOST planes (sample (0, 0), padded by pad_l / pad_c samples), pred_* = HOST blocks (pred[0][Y_C / U_C / V_C]).  Synchronous.  What the Main encoder's
 * by-name calls of xeve_affine_mc are routed to in situ (oracle/ref_shim_affine.c). */
int xeve_hip_affine_mc_host(int x, int y, int pic_w, int pic_h, int w, int h, const int8_t refi[2], const int16_t mv[2][3][2], const xeve_hip_refpic *refp, int num_refp0,
                            int num_refp1, int s_l, int s_c, int pad_l, int pad_c, xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, int vertex_num,
                            int bit_depth);

/* ------------------------------------------------------------------------------------------- */
/* Main profile: the affine gradient search ("affine MC + gradient ME").   */
/* reference: pinter_affine_me_gradient (src_main/xevem_pinter.c:4290-4501; pi->fn_affine_me,      */
/* src_base/xeve_type.h:448) = luma affine compensation (xeve_affine_mc_l, xevem_mc.c:1532-1669),  */
/* SATD + vector bits (get_affine_mv_bits :4257-4288), then per round the error, the prediction's  */
/* Sobel derivatives, the normal equations (xevem_func_aff_h_sobel_flt / _v_sobel_flt /            */
/* _eq_coef_comp), solve_equal (:4213-4255) in double and the control points' rounded update;      */
/* 7 / 5 rounds (uni / bi), two fewer with three control points; the best vectors are kept.        */
/* ------------------------------------------------------------------------------------------- */
typedef struct xeve_hip_affine_me_job {
    int32_t  x, y;           /* CU position in luma samples */
    int16_t  mvp[3][2];      /* the predictor's control points (the vector bits are counted against them) */
    int16_t  mv[3][2];       /* in: the start vectors; out: the best ones found (vertex_num of them) */
    int8_t   refi, list;     /* the reference picture: refp[refi * 2 + list] */
    int8_t   bi;             /* 1: the original is the job's block of org_bi (pi->org_bi: 2 * org - the other list's prediction), SATD >> 1, + mot_bits_other */
    int8_t   vertex_num;     /* 2 | 3 */
    int32_t  mot_bits_other; /* pi->mot_bits[1 - list] */
    uint32_t cost;           /* out: the function's value, cost_best - MV_COST(best_bits) */
} xeve_hip_affine_me_job;
/* Every search of ONE CU size in one call (a workgroup per search, all rounds inside the kernel).  refp: HOST array [refi * 2 + list] (ntab0 / ntab1 entries per list) of
 * DEVICE luma planes (sample (0, 0); u / v are not read), padded like the reference's; org: DEVICE original luma plane (sample (0, 0)), pitch s_org; org_bi: DEVICE
 * [njobs][h * w] 16-bit blocks, read for the jobs that have bi set (may be NULL when none has); jobs: DEVICE, read and written.  lambda_mv = pi->lambda_mv; num_refp0 / 1 =
 * ctx->rpm.num_refp[list] (the reference-index bits).  w, h: powers of two, 16 .. 128 (the encoder searches affine vectors for CUs of 16 x 16 and more, :5516). */
int xeve_hip_affine_me_jobs(const xeve_hip_refpic *refp, int ntab0, int ntab1, int s_l, int pic_w, int pic_h, const xeve_hip_pel *org, int s_org, const int16_t *org_bi,
                            xeve_hip_affine_me_job *jobs, int njobs, int w, int h, int bit_depth, uint32_t lambda_mv, int num_refp0, int num_refp1, void *stream);
/* Host-memory form of ONE pinter_affine_me_gradient call: ref_y = HOST luma plane (sample (0, 0)) of refp[refi][list].pic, padded by pad_l samples; org = the CU's first
 * original sample with its pitch (bi: pi->org_bi, pitch w); mv in / out; *cost = the function's value.  Synchronous.  What the Main encoder's pi->fn_affine_me is bound to in
 * situ (oracle/ref_shim_affine.c). */
int xeve_hip_affine_me_host(int x, int y, int pic_w, int pic_h, int w, int h, int refi, int list, const int16_t mvp[3][2], int16_t mv[3][2], int bi, int vertex_num,
                            const xeve_hip_pel *ref_y, int s_l, int pad_l, const int16_t *org, int s_org, int bit_depth, uint32_t lambda_mv, int num_refp, int mot_bits_other,
                            uint32_t *cost);

#ifdef __cplusplus
}
#endif
#endif /* XEVE_HIP_H */
