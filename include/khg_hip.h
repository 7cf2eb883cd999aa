/* include/khg_hip.h -- C-ABI of the MI355X (gfx950) HMM-GMM EM hot path.
 *
 * Drop-in boundary for csukuangfj/kaldi-hmm-gmm's align + acc-stats + M-step path.  Every
 * entry point names the reference interface it replaces (paths relative to
 * /root/reference/kaldi-hmm-gmm/, "csrc/" = the core library, "python/csrc/" = the pybind11
 * layer a maintainer would bind these from).
 *
 * Conventions: plain pointers and sizes only; `_h` = host pointer, `_d` = device pointer;
 * every function returns 0 on success or a negative KHG_E_* code, with the message available
 * from khg_last_error() (thread-local).  Reference KHG_ERR / KHG_ASSERT (csrc/log.h:46-83,
 * std::runtime_error) map to KHG_E_RUNTIME.  Handles are opaque; work is stream-ordered on the
 * context's HIP stream; there is no hidden CPU fallback: without a usable GPU every compute
 * entry point fails with KHG_E_HIP.
 */
#ifndef KHG_HIP_H_
#define KHG_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KHG_OK 0
#define KHG_E_ARG (-1)      /* invalid argument / shape mismatch                              */
#define KHG_E_HIP (-2)      /* HIP runtime error (no device, OOM, launch failure)             */
#define KHG_E_RUNTIME (-3)  /* the reference would throw std::runtime_error here              */
#define KHG_E_UNSUPPORTED (-4)
/* largest feature dimension: a 64-frame chunk of rows must fit LDS next to a pdf's posteriors.  D <= 80 runs the MFMA
 * kernels; 80 < D <= KHG_MAX_DIM the vector-ALU forms of K1 and K3 (correct, not tuned). */
#define KHG_MAX_DIM 512

typedef struct khg_ctx khg_ctx;
typedef struct khg_model khg_model;
typedef struct khg_tm khg_tm;
typedef struct khg_utts khg_utts;
typedef struct khg_accs khg_accs;

const char *khg_last_error(void);
int khg_version(void);

/* ---- per-device context ------------------------------------------------------------- */
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream) or NULL to let the
 * context create its own non-blocking stream. */
int khg_ctx_create(int device, void *stream, khg_ctx **out);
/* Waits for the context's streams.  Handles made on the context (models, tables, utterance sets, statistics blocks) may be destroyed
 * after it; their device memory stays valid until then (a statistics block can still be read through another context), and every
 * entry point that is given the destroyed context returns KHG_E_ARG. */
int khg_ctx_destroy(khg_ctx *ctx);
/* Waits for the stream and reports kernel-side errors deferred by the asynchronous entry points
 * (khg_loglikes, khg_align without host outputs, khg_acc_stats): KHG_E_RUNTIME where the reference
 * would have thrown (NaN/Inf log-likelihood, pdf-id out of range). */
int khg_ctx_sync(khg_ctx *ctx);
/* Measurement aid: when on, every kernel launch is bracketed by HIP events on the
 * context's stream; khg_ctx_get_timings drains (name, ms) pairs, names '\n'-separated. */
int khg_ctx_set_timing(khg_ctx *ctx, int on);
int khg_ctx_get_timings(khg_ctx *ctx, char *names, int64_t names_cap, float *ms, int32_t cap,
                        int32_t *n_out);

/* Which arithmetic K1 (khg_loglikes*) runs in.  All forms evaluate decodable-am-diag-gmm.cc:55-61 to fp32 accuracy
 * (|error| <= 1e-5 + 1e-6 B against an fp64 evaluation, B = |gconst| + sum |M x| + sum |V x^2| / 2; the reference's own
 * Eigen gemv fixes no summation order either):
 *   KHG_K1_F16X2S    (what AUTO selects; csrc/khg_k1_f16x2s.hip.inc) both operands rescaled per contraction index by an exact
 *                    power of two (feature columns and the largest weight column peak in [2^14, 2^15)) and split into two fp16
 *                    pieces v1 + v2, v1 = fp16(v), v2 = fp16(v - v1): 11 + 11 significant bits, |v - (v1 + v2)| <=
 *                    max(2^-23 |v|, 2^-25) in scaled units; the three partial products w1 x1, w1 x2, w2 x1 on
 *                    v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator (the dropped w2 x2 is <= 2^-22 of the term: worst case
 *                    2^-21 per term plus an absolute floor, measured 2.1e-7 B against fp64).  Used while the split forms'
 *                    common domain holds (max |gconst| + sum_k max |w_k| max |x_k| <= 2^28) and the floor, summed at the column
 *                    maxima, stays <= 2e-6; otherwise khg_loglikes runs KHG_K1_F16X2, then KHG_K1_FP32_PDF / _UTT, by itself;
 *   KHG_K1_F16X2     the same three products with the residual pieces pre-scaled by 2^11 (v = v1 + v2 2^-11: no subnormal
 *                    residuals, |v - (v1 + v2 2^-11)| <= 2^-23 |v|) and two accumulators (main + cross, combined by one fma per
 *                    Gaussian): worst case 2^-21 per term, measured error BELOW the fp32 chain's (profiles/r2_probe_f16x2.txt);
 *                    same common domain;
 *   (value 1, KHG_K1_BF16X3 of rounds 2-5 -- three exact bf16 pieces, six products -- was removed in round 6: KHG_E_ARG)
 *   KHG_K1_FP32_PDF / KHG_K1_FP32_UTT   fp32 MFMA (v_mfma_f32_16x16x4_f32), pdf-major / utterance-major tiling: bit for bit the
 *                    per-Gaussian chain s = gconst; s = fmaf(M[d], x[d], s) ...; s = fmaf(-V[d]/2, x[d]^2, s) ... in k order.
 * The environment variable KHG_K1 = f16x2s | f16x2 | pdf | utt seeds the setting at khg_ctx_create (A/B runs). */
#define KHG_K1_AUTO 0
#define KHG_K1_FP32_PDF 2
#define KHG_K1_FP32_UTT 3
#define KHG_K1_F16X2 4
#define KHG_K1_F16X2S 5
int khg_ctx_set_k1_form(khg_ctx *ctx, int form);     /* = khg_ctx_set_option(ctx, KHG_OPT_K1_FORM, form) */

/* Every switch of the library that is not an argument of an entry point: kernel forms and experiment knobs (what the reference
 * has no counterpart for -- its kernels are Eigen expressions).  Values are validated (KHG_E_ARG).  An option takes effect at the
 * next call that plans or launches the kernel it names; plans already built for an utterance set (chunk order, K1P slices) are
 * kept.  The environment variables in brackets only seed the defaults, once, at khg_ctx_create (A/B runs of an unmodified caller). */
#define KHG_OPT_K1_FORM 0        /* KHG_K1_* above                                                              [KHG_K1=f16x2s|f16x2|pdf|utt] */
#define KHG_OPT_K1_ORDER 1       /* launch order of K1 workgroups: 0 frame tiles x pdfs descending, 1 utterance order, 2 ascending,
                                    3 frame tiles descending, 4 the chunks of one utterance eight positions apart (one XCD)  [KHG_K1_ORDER=desc|none|asc|tiles|xcd] */
#define KHG_OPT_K1_NF 2          /* fp32 utterance-major K1: 16-frame tiles per wave at D <= 40 (0 = 6, or 5)     [KHG_K1_NF] */
#define KHG_OPT_K1P_TS 3         /* fp32 pdf-major K1: tiles per workgroup slice (default 1024)                  [KHG_K1P_TS] */
#define KHG_OPT_K1_INTERLEAVE 4  /* fp32 utterance-major K1: frame tiles dealt round-robin (-1 auto, 0, 1)        [KHG_K1_INTERLEAVE] */
#define KHG_OPT_K1_DBG 5         /* experiment bit mask (bits 1-8: the tile-major split forms, results may be WRONG; 16: no packing of
                                    small pdfs; 32 / 64: K1s band form without shifted tiles / 16-frame shift only)  [KHG_K1B_DBG] */
#define KHG_OPT_K2_INORDER 6     /* 1: K2 workgroups in utterance order instead of longest first                  [KHG_K2_INORDER] */
#define KHG_OPT_K2_KS 7          /* states per thread on K2's register-resident path: 0 auto, 2, 4; 3 = the general three-slot kernel also where the two-slot one applies [KHG_K2_KS] */
#define KHG_OPT_K2_SERIAL 8      /* 1: the one-lane order-faithful decoder also where the wave form applies;
                                    2: the wave form with its graph tables in HBM scratch also where they fit LDS;
                                    3: the general wave form also where the chain form (no epsilon arcs, <= 1000 states,
                                       out-degree <= 4) applies                                                      [KHG_K2_SERIAL] */
#define KHG_OPT_K2_PROF 9        /* 1: per-utterance cycle stamps of K2 to stderr                                 [KHG_K2_PROF] */
#define KHG_OPT_K3_BUCKET 10     /* frames by pdf: 0 stable radix sort of (pdf, frame) pairs (rocPRIM; reproducible sums), 1 atomic cursor scatter, 2 the library's own stable counting sort (same order as 0, slower) [KHG_K3_BUCKET=sort|atomic|count] */
#define KHG_OPT_K3_FORM 11       /* 0 auto, 1 the chunk-per-block MFMA form for every shape, 2 the VALU form      [KHG_K3_FORM=block, KHG_K3_VALU=1] */
#define KHG_OPT_K3_PHASE_B 12    /* gamma . x: 0 on the fp64 matrix pipe (exact products), 1 as 0 (the fp32-pipe form of rounds 3-5 was removed in round 6), 2 (DEFAULT) on the fp16 matrix cores (operands split into two fp16 pieces, 32-frame fp32 partial sums added in fp64; pdfs of 33..64 Gaussians, D <= 40, else as 0) [KHG_K3_PHASEB=f32|f16] */
#define KHG_OPT_K3_NY 13         /* workgroups per pdf in K3 (0 auto)                                             [KHG_K3_NY] */
#define KHG_OPT_DEBUG 14         /* 1: planning statistics to stderr                                              [KHG_DEBUG] */
#define KHG_OPT_K3_PHASE_A 15    /* per-Gaussian log-likelihoods of K3's wave form: 0 on the fp16 matrix cores in K1's f16x2s arithmetic where the model-derived scales hold, 1 the fp32 MFMA chain [KHG_K3_PHASEA=f32] */
#define KHG_OPT_K2_SPLIT 16      /* an asynchronous khg_align (no host outputs) on a set of > 64 utterances lets the order-faithful decoders write to a
                                    second alignment buffer, so that khg_acc_stats can accumulate the certified utterances while they still
                                    run and add the others in a second pass: 0 (DEFAULT) on, 1 off                       [KHG_K2_SPLIT=off] */
#define KHG_OPT_COUNT 17
/* Read-only figures (khg_ctx_get_option only): the per-call scratch block behind small utterance sets (DESIGN.md "per-utterance calls"). */
#define KHG_INFO_SCRATCH_BYTES 100   /* bytes of the block in use (its top), 0 before the first small set */
#define KHG_INFO_SCRATCH_BLOCKS 101  /* live allocations inside it */
int khg_ctx_set_option(khg_ctx *ctx, int option, int value);
int khg_ctx_get_option(const khg_ctx *ctx, int option, int *value);

/* ---- acoustic model ------------------------------------------------------------------- */
/* AmDiagGmm (csrc/am-diag-gmm.h:96) as flat ragged arrays: pdf p owns Gaussians
 * [gauss_off[p], gauss_off[p+1]) with DiagGmm's exponential-form parameters
 * (csrc/diag-gmm.h:243-256).  gconsts must be valid (khg_compute_gconsts).  Uploads the
 * MFMA tile image used by K1 and the row-major copy used by K3. */
int khg_model_create(khg_ctx *ctx, int32_t num_pdfs, int32_t dim, const int32_t *gauss_off_h,
                     const float *gconsts_h, const float *means_invvars_h,
                     const float *inv_vars_h, khg_model **out);
int khg_model_destroy(khg_model *m);
/* Measurement / consistency aid: drops what the handle caches PER PARAMETER VERSION -- the fp16 / bf16 K1 images, the BAND form's
 * per-pdf upper bounds, the column maxima and scale exponents -- as every in-place update (khg_model_mle_update, _split, _merge,
 * _scale_weights) does, without touching the parameters: the next khg_loglikes / khg_acc_stats derive them again.  One EM iteration of
 * the reference changes the parameters once (scripts/gmm_est.py:8-96), so a benchmark step that is to pay what a real iteration
 * pays calls this once per step (bench.py). */
int khg_model_invalidate(khg_model *m);

/* ---- transition information ------------------------------------------------------------ */
/* TransitionInformation::TransitionIdToPdf table (csrc/transition-information.h:71-73,
 * csrc/transition-model.cc:278-302): id2pdf_h[0..num_tids], entry 0 unused. */
int khg_tm_create(khg_ctx *ctx, int32_t num_tids, const int32_t *id2pdf_h, khg_tm **out);
/* AddTransitionProbs (csrc/hmm-utils.cc:465-493): trans_cost_h[tid] = -GetScaledTransitionLogProb
 * (:442-463), added to every arc carrying that tid when khg_align runs (the resident graphs keep
 * their base weights, as scripts/gmm_align_compiled.py:36-41 does on a copy per call).
 * NULL resets to "no transition probs added". */
int khg_tm_set_trans_cost(khg_tm *tm, const float *trans_cost_h);
int khg_tm_destroy(khg_tm *tm);

/* ---- utterances: features + decoding graphs, resident in HBM --------------------------- */
/* feats: [frame_off[n_utt]][dim] float32 row-major, either host (feats_h) or already on the
 * device (feats_d, borrowed: must outlive the handle).  Graphs: fst::VectorFst<StdArc> per
 * utterance (the `fst` argument of python/csrc/decoder-wrappers.cc:25-47) concatenated as CSR
 * by source state: utterance u owns states [state_off[u], state_off[u+1]); global state s owns
 * arcs [arc_off[s], arc_off[s+1]); nextstate is utterance-local; start_h[u] = -1 for an empty
 * FST; final_h[s] = +inf for non-final (TropicalWeight::Zero()).  Pass n_states_total = 0 and
 * NULL graph arrays for a features-only set (log-likes / acc-stats without alignment).
 * The reference's per-utterance calls (scripts/gmm_align_compiled.py:36-79, gmm_acc_stats_ali.py:46-58) become one set of ONE
 * utterance per call: a set of <= 16 utterances (<= 16 384 frames) takes all its device scratch from a per-context arena mirrored in
 * pinned host memory -- no hipMalloc / hipFree, its tables reach the device in one staged copy, khg_align's results come back in
 * one -- so create / khg_loglikes_reachable / khg_align / destroy of a 300-frame utterance is ~0.35 ms at 5000 x 64 x 40. */
int khg_utts_create(khg_ctx *ctx, const khg_tm *tm, int32_t n_utt, int32_t dim,
                    const int64_t *frame_off_h, const float *feats_h, const float *feats_d,
                    const int64_t *state_off_h, const int32_t *start_h, const int64_t *arc_off_h,
                    const int32_t *ilabel_h, const int32_t *olabel_h, const float *weight_h,
                    const int32_t *nextstate_h, const float *final_h, khg_utts **out);
int khg_utts_destroy(khg_utts *u);
/* number of distinct pdfs on each utterance's graph, and the list itself (sorted) */
int khg_utts_num_pdfs(const khg_utts *u, int64_t *pdf_off_h /* [n_utt+1] */);
int khg_utts_pdfs(const khg_utts *u, int32_t *pdfs_h /* [pdf_off[n_utt]] */);
/* per listed pdf: the first frame a decoder token can read it at (fewest emitting arcs from the start state
 * to an arc carrying it; INT32_MAX if never; 0 for sets without graphs) -- what khg_loglikes_reachable uses */
int khg_utts_pdf_first(const khg_utts *u, int32_t *first_h /* [pdf_off[n_utt]] */);

/* ---- K1: log-likelihoods --------------------------------------------------------------- */
/* DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased (csrc/decodable-am-diag-gmm.cc:29-71) for
 * every (frame, pdf on the utterance's graph); DiagGmm::LogLikelihood (csrc/diag-gmm.cc:150-165)
 * when a pdf list is given explicitly.  Result layout per utterance: ll[j * tpad + t],
 * tpad = T rounded up to 32, j = index into the utterance's pdf list.  KHG_E_RUNTIME if any
 * value is NaN/Inf (the reference throws, :63-65). */
int khg_loglikes(khg_ctx *ctx, const khg_model *m, khg_utts *u);
/* Same, restricted to the (frame, pdf) cells the decoder can read: DecodableAmDiagGmmScaled only
 * evaluates LogLikelihood(frame, tid) for tokens that exist (csrc/faster-decoder.cc:208), and no token
 * can sit in a state before as many frames as the fewest emitting arcs from the start state lead to
 * it.  Cells of a pdf before its first readable frame are unspecified (whole tiles in front of it are
 * left untouched, and a pdf's tiles may start at that very frame); alignments are identical to
 * khg_loglikes + khg_align.  Sets without graphs: same as khg_loglikes. */
int khg_loglikes_reachable(khg_ctx *ctx, const khg_model *m, khg_utts *u);
/* The BAND form: additionally leaves out what only tokens that can no longer reach a final state read -- a (pdf, 32-frame tile)
 * past the last frame at which an arc carrying the pdf still leads to a final state by the utterance's end (fewest emitting arcs
 * to a final state, khg_utts_pdf_last).  Those cells are FILLED with an upper bound of the pdf's log-likelihood, so the exact DP
 * of khg_align sees such tokens at costs no higher than the reference decoder would -- its beam certificate stays sound and the
 * best path is untouched.  An utterance whose certificate fails is recomputed without the band by khg_align itself before the
 * order-faithful decoder reads it: EVERY khg_align on these scores needs the model handle alive and at the parameter version the scores
 * were computed with -- a handle destroyed or updated in between makes khg_align return KHG_E_ARG (checked by handle serial and
 * version, never by dereferencing a stale pointer); a handle whose fp16 image was merely re-packed for another set's feature
 * exponents makes khg_align score the set again first.  Alignments are identical to khg_loglikes + khg_align at any beam.  Worth it when the beam is wide (few certificates fail): ~21 % fewer cells at
 * the benchmark's shape.  Cells of a pdf BEFORE its first readable frame are unspecified (as with khg_loglikes_reachable, here to
 * the frame: a band's 32-frame tiles may start at that frame instead of on the 32-frame grid).  Default K1 form only (f16x2s, pdfs
 * of more than 16 Gaussians); anything else: khg_loglikes_reachable. */
int khg_loglikes_band(khg_ctx *ctx, const khg_model *m, khg_utts *u);
/* per listed pdf: the last frame at which an arc carrying it can still lead to a final state by the utterance's end (-1: never;
 * INT32_MAX for sets without graphs) */
int khg_utts_pdf_last(const khg_utts *u, int32_t *last_h /* [pdf_off[n_utt]] */);
/* total floats of the resident ll buffer and per-utterance offsets [n_utt+1] */
int khg_loglikes_layout(const khg_utts *u, int64_t *ll_off_h, int64_t *total);
int khg_loglikes_download(khg_ctx *ctx, const khg_utts *u, float *ll_h);
/* test hook: overwrite the resident ll buffer (lets K2 be checked bit-exactly on given scores) */
int khg_loglikes_upload(khg_ctx *ctx, khg_utts *u, const float *ll_h);
/* features-only sets: give every utterance the same explicit pdf list */
int khg_utts_set_pdf_list(khg_utts *u, int32_t n, const int32_t *pdfs_h);
/* Borrowed device features (feats_d of khg_utts_create) are otherwise IMMUTABLE for the life of the handle: the default K1 keeps
 * per-set data derived from them (column maxima, fp16 planes packed once).  A caller that rewrites them in place calls this
 * before the next khg_loglikes; K3 and the fp32 K1 forms read feats_d live. */
int khg_utts_features_changed(khg_utts *u);

/* ---- K2: Viterbi forced alignment ------------------------------------------------------ */
typedef struct {
  float beam;        /* AlignConfig (csrc/decoder-wrappers.h:23-37): 200 */
  float retry_beam;  /* 0 */
  int32_t careful;   /* 0; careful alignment = graphs passed through khg_careful_graph by the caller before khg_utts_create */
  float acoustic_scale;
  /* FasterDecoderOptions (csrc/faster-decoder.h:24-49); AlignUtteranceWrapper keeps defaults */
  int32_t max_active; /* INT32_MAX */
  int32_t min_active; /* 20 */
  float beam_delta;   /* 0.5 */
  float hash_ratio;   /* 2.0 */
  /* divisor of `like` (decoder-wrappers.cc:95) when it differs from the scale the scores are multiplied by: the reference scales
   * scores inside the decodable (its own `scale`) and divides `like` by AlignUtteranceWrapper's acoustic_scale argument.
   * 0 = acoustic_scale (the scripts pass the same value for both). */
  float like_scale;
} khg_align_config;
void khg_align_config_default(khg_align_config *c);

/* per-utterance status bits */
#define KHG_ALIGN_DONE 0
#define KHG_ALIGN_ERROR 1     /* num_error++ (empty graph / no final state reached)            */
#define KHG_ALIGN_RETRIED 2   /* num_retried++                                                 */
#define KHG_ALIGN_EXACT_DP 4  /* info: produced by the exact-DP kernel under a beam certificate */
#define KHG_ALIGN_FALLBACK 8  /* info: produced by the order-faithful FasterDecoder kernel     */

/* ModifyGraphForCarefulAlignment (csrc/decoder-wrappers.cc:111-140: fst := Concat(fst, fst_rhs), fst_rhs = a copy of fst
 * without final weights, entered through a new final start state by an epsilon arc) on ONE graph in the CSR-by-source
 * layout of khg_utts_create (host arrays; nextstate graph-local).  OpenFst's Concat semantics: every final state s of the
 * left copy loses its final weight w and gains an epsilon arc (0:0 / w) to the right copy's start, appended after its own
 * arcs.  Result: 2 S + 1 states (left copy 0..S-1, right copy S..2S-1, the pre-initial state 2S, final with weight 0),
 * 2 A + 1 + (#final states) arcs; the caller sizes the output arrays for that ([2S+2] arc_off, [2A+1+S] arc arrays,
 * [2S+1] final).  An empty graph (num_states == 0) is returned unchanged.  Host only. */
int khg_careful_graph(int32_t num_states, int32_t start, const int64_t *arc_off_h, const int32_t *ilabel_h,
                      const int32_t *olabel_h, const float *weight_h, const int32_t *nextstate_h, const float *final_h,
                      int32_t *out_num_states, int32_t *out_start, int64_t *out_arc_off_h, int32_t *out_ilabel_h,
                      int32_t *out_olabel_h, float *out_weight_h, int32_t *out_nextstate_h, float *out_final_h);

/* AlignUtteranceWrapper (csrc/decoder-wrappers.cc:16-108) + FasterDecoder (csrc/faster-decoder.cc)
 * + DecodableAmDiagGmmScaled (csrc/decodable-am-diag-gmm.h:83-103) for every utterance of the
 * set.  Requires khg_loglikes() (or khg_loglikes_upload) first.  Outputs (host, may be NULL):
 *   ali_h[frame_off[n_utt]]  transition-ids, 0 for failed utterances
 *   words_h / words_off_h[n_utt+1]: olabels != 0 along the best path (words_cap = capacity)
 *   like_h[n_utt]   float `like` of decoder-wrappers.cc:95
 *   status_h[n_utt] KHG_ALIGN_* bits.
 * The alignment also stays resident on the device for khg_acc_stats.  With every host output NULL the call is asynchronous: the exact
 * DP runs on the context's stream, the order-faithful decoders for the utterances it could not certify on a side stream; on a set of
 * more than 64 utterances those write to a second alignment buffer that the next consumer merges (KHG_OPT_K2_SPLIT). */
int khg_align(khg_ctx *ctx, const khg_tm *tm, khg_utts *u, const khg_align_config *cfg,
              int32_t *ali_h, int32_t *words_h, int64_t *words_off_h, int64_t words_cap,
              float *like_h, int32_t *status_h);
/* replace the resident alignment (e.g. an initial equal-align, egs/yesno/train.py:86-108) */
int khg_ali_upload(khg_ctx *ctx, khg_utts *u, const int32_t *ali_h);
/* the resident alignment back (0 on the frames of utterances that failed to align) */
int khg_ali_download(khg_ctx *ctx, khg_utts *u, int32_t *ali_h);

/* ---- K3: sufficient statistics ---------------------------------------------------------- */
/* AccumAmDiagGmm (csrc/mle-am-diag-gmm.h:93-96) + transition stats (csrc/transition-model.h:176-189)
 * as ONE contiguous fp64 device buffer (a single RCCL all-reduce sums it across GPUs =
 * AccumAmDiagGmm::Add, csrc/mle-am-diag-gmm.cc:119-128):
 *   [ occ: sumG | mean_acc: sumG*dim | var_acc: sumG*dim | trans_acc: num_tids+1 |
 *     total_frames, total_log_like, 6 spare ]                                                  */
int khg_accs_create(khg_ctx *ctx, const khg_model *m, const khg_tm *tm, khg_accs **out);
int khg_accs_destroy(khg_accs *a);
int khg_accs_zero(khg_ctx *ctx, khg_accs *a);
int khg_accs_size(const khg_accs *a, int64_t *num_doubles);
int khg_accs_device_ptr(const khg_accs *a, void **ptr_d);
int khg_accs_download(khg_ctx *ctx, const khg_accs *a, double *buf_h);
int khg_accs_upload(khg_ctx *ctx, khg_accs *a, const double *buf_h);

/* scripts/gmm_acc_stats_ali.py:46-56 for every frame of every utterance with a resident
 * alignment: AccumAmDiagGmm::AccumulateForGmm (csrc/mle-am-diag-gmm.cc:41-52) ->
 * AccumDiagGmm::AccumulateFromDiag/FromPosteriors (csrc/mle-diag-gmm.cc:123-158) ->
 * DiagGmm::ComponentPosteriors (csrc/diag-gmm.cc:368-392), plus tacc[tid] += 1.
 * Asynchronous on the context's stream.  Right behind an asynchronous khg_align on a large set the call waits ONCE for the exact DP
 * (not for the order-faithful decoders, not for its own kernels) to learn whether any utterance was left to those decoders; if so
 * the certified utterances are accumulated at once and the others in a second pass when the decoders are done (the statistics are
 * additive, csrc/mle-am-diag-gmm.cc:41-52; KHG_OPT_K2_SPLIT = 1: one pass after the decoders). */
int khg_acc_stats(khg_ctx *ctx, const khg_model *m, const khg_tm *tm, khg_utts *u, float weight,
                  khg_accs *a);

/* ---- C1: cross-GPU sum of the accumulator block (one process per GPU) -------------------- */
/* AccumAmDiagGmm::Add across jobs (csrc/mle-am-diag-gmm.cc:119-128; what Kaldi's gmm-sum-accs does on
 * files): ONE in-place ncclAllReduce(sum, fp64) of the whole block [occ | mean_acc | var_acc | trans_acc |
 * scalars], enqueued on the context's stream right behind khg_acc_stats -- no host synchronisation
 * between K3, the exchange and khg_model_mle_update.  `comm` is an RCCL ncclComm_t for the context's
 * device: the caller's own (ncclCommInitRank) or one made with khg_comm_create.  The library does not
 * link RCCL: it binds ncclAllReduce & co. from the librccl already loaded into the process (torch's, the
 * caller's) or loads librccl.so.1 itself.  comm == NULL: a one-rank job, nothing to exchange. */
int khg_accs_allreduce(khg_ctx *ctx, khg_accs *a, void *comm);
/* The same sum for the accumulator rows of pdfs [first_pdf, first_pdf + n_pdf) only -- the three contiguous pieces occ / mean_acc /
 * var_acc of their Gaussians, one RCCL group -- or, with first_pdf < 0, for the transition counts and scalars behind them.  The
 * ranges of a partition of the pdfs plus the tail add up to khg_accs_allreduce. */
int khg_accs_allreduce_range(khg_ctx *ctx, khg_accs *a, const khg_model *m, int32_t first_pdf, int32_t n_pdf, void *comm);
/* khg_acc_stats with C1 PIPELINED behind it: the pdfs are cut into `nparts` ranges (<= 0: 4); while the accumulate kernels of
 * range i + 1 run on the context's stream, the rows of range i are all-reduced on a second stream of the context; the context's
 * stream waits for the last piece, so the caller goes on exactly as after khg_acc_stats + khg_accs_allreduce.  For the LAST
 * khg_acc_stats of a pass only (the block must be complete); comm == NULL: plain khg_acc_stats.  On two ranks the result is
 * bit-identical to the unpipelined exchange (a + b has one order); on more, each form is reproducible run to run. */
int khg_acc_stats_reduce(khg_ctx *ctx, const khg_model *m, const khg_tm *tm, khg_utts *u, float weight, khg_accs *a,
                         void *comm, int32_t nparts);
/* BASELINE.json configs[4] "fp32 stats vs CPU tolerance check": the same exchange with the block
 * rounded to fp32 for the wire (convert -> ncclAllReduce(sum, fp32) -> widen back into the block):
 * half the xGMI bytes, per-rank partial sums lose their low 29 bits.  comm == NULL: only the rounding. */
int khg_accs_allreduce_f32(khg_ctx *ctx, khg_accs *a, void *comm);
/* communicator helpers for callers without their own RCCL plumbing: rank 0 calls khg_comm_unique_id,
 * ships the 128 bytes to the other ranks by any means (a file, MPI, torch.distributed's store), every
 * rank then calls khg_comm_create (collective; blocks until all `nranks` ranks have called it). */
#define KHG_COMM_ID_BYTES 128
int khg_comm_unique_id(void *id_out /* [KHG_COMM_ID_BYTES] */);
int khg_comm_create(khg_ctx *ctx, int32_t nranks, int32_t rank, const void *id, void **comm_out);
int khg_comm_destroy(void *comm);
/* what RCCL reports for a communicator (any of the outputs may be NULL): ncclCommCount, ncclCommUserRank, ncclGetVersion
 * (comm == NULL: only the version) -- lets a multi-GPU run state on its own record how many ranks the collective really spanned */
int khg_comm_info(void *comm, int32_t *nranks, int32_t *rank, int32_t *version);

/* ---- host-side M-step and helpers (no GPU needed) --------------------------------------- */
/* DiagGmm::ComputeGconsts (csrc/diag-gmm.cc:103-147) for a ragged model; num_bad_out may be NULL */
int khg_compute_gconsts(int32_t num_pdfs, int32_t dim, const int32_t *gauss_off,
                        const float *weights, const float *inv_vars, const float *means_invvars,
                        float *gconsts, int32_t *num_bad_out);

typedef struct {
  float min_gaussian_weight;          /* MleDiagGmmOptions (csrc/mle-diag-gmm.h:23-45): 1e-5 */
  float min_gaussian_occupancy;       /* 10 */
  double min_variance;                /* 1e-3 */
  int32_t remove_low_count_gaussians; /* 1 */
  /* variance_floor_vector (csrc/mle-diag-gmm.h:26-28, used at csrc/mle-diag-gmm.cc:311-322): per-dimension floor, host
   * pointer to `dim` doubles, NULL = not supplied (then min_variance floors every dimension) */
  const double *variance_floor_vector;
} khg_mle_options;
void khg_mle_options_default(khg_mle_options *o);

/* MleAmDiagGmmUpdate (csrc/mle-am-diag-gmm.cc:153-202) over flat arrays.  In: accumulator
 * arrays laid out like the model (gauss_off); in/out: weights, gconsts, means_invvars, inv_vars
 * compacted in place when Gaussians are removed; out: new_gauss_off[num_pdfs+1],
 * objf_change, count (floats, as the reference returns them). */
int khg_mle_am_diag_gmm_update(const khg_mle_options *o, int32_t num_pdfs, int32_t dim,
                               const int32_t *gauss_off, const double *occ, const double *mean_acc,
                               const double *var_acc, uint16_t acc_flags, uint16_t flags,
                               float *weights, float *gconsts, float *means_invvars,
                               float *inv_vars, int32_t *new_gauss_off, float *objf_change,
                               float *count, int32_t *floored_elems, int32_t *floored_gauss,
                               int32_t *removed);

/* DiagGmm::Merge (csrc/diag-gmm.cc:557-759, MergedComponentsLogdet :761-778) on one pdf's flat arrays: greedy merging of
 * the pair whose union loses the least likelihood, down to target_components (1: the global mean and variance); arrays are
 * compacted in place, gconsts recomputed; history_out (may be NULL) receives the merged pairs (kept, removed) in order --
 * what python/csrc/diag-gmm.cc:79-85 returns from DiagGmm.merge.  AmDiagGmm::MergeByCount (csrc/am-diag-gmm.cc:91-108) =
 * this per pdf with the targets of GetSplitTargets.  Host only. */
int khg_diag_gmm_merge(int32_t *num_gauss, int32_t dim, int32_t target_components, float *weights, float *gconsts,
                       float *means_invvars, float *inv_vars, int32_t *history_out, int32_t *num_history_out);

/* ---- K4: the same M-step on the device ---------------------------------- */
/* MleAmDiagGmmUpdate (csrc/mle-am-diag-gmm.cc:153-202; per pdf MleDiagGmmUpdate,
 * csrc/mle-diag-gmm.cc:243-390, DiagGmmNormal csrc/diag-gmm-normal.cc:14-48, ComputeGconsts
 * csrc/diag-gmm.cc:103-147, RemoveComponents :853-938) run on the (all-reduced) accumulators where K3
 * left them: no 205 MB accumulator download, no host update, no parameter upload.  The model handle is
 * updated IN PLACE (row-major parameters, the K1 tile image, gauss_off when Gaussians were removed --
 * then call khg_accs_relayout before the next khg_acc_stats).  weights, inv_vars and means_invvars are
 * bit-identical to khg_mle_am_diag_gmm_update; gconsts / objf_change go through logf and may differ from
 * the host's in the last place.  The mixture weights (which khg_model_create does not take; K1-K3 only
 * need gconsts) must have been set with khg_model_set_weights. */
int khg_model_set_weights(khg_ctx *ctx, khg_model *m, const float *weights_h);
int khg_model_mle_update(khg_ctx *ctx, khg_model *m, const khg_accs *a, const khg_mle_options *o,
                         uint16_t flags, float *objf_change, float *count, int32_t *floored_elems,
                         int32_t *floored_gauss, int32_t *removed);
/* The SHARDED M-step of a multi-GPU job (MleAmDiagGmmUpdate is independent per pdf, csrc/mle-am-diag-gmm.cc:153-202): rank r of
 * nranks owns the pdfs [P r / nranks, P (r + 1) / nranks).  `a` holds the rank's LOCAL sums (no khg_accs_allreduce before): its
 * mean / variance rows are ncclReduce'd to their owners by pdf range, the occupancies all-reduced (every rank's mixing-up targets need
 * them; the transition counts and scalars are NOT touched: khg_accs_allreduce_range with first_pdf < 0), every rank updates its own
 * pdfs, the rewritten rows and per-pdf results are ncclBroadcast from their owners, and every rank finishes on the complete model
 * (compaction when Gaussians were removed, images): bit-identical to khg_model_mle_update on the all-reduced block on two ranks
 * (and with one), (nranks - 1) / nranks x (accumulator + parameter bytes) on the wire per rank instead of 2 (nranks - 1) / nranks x
 * accumulator bytes.  comm == NULL: khg_model_mle_update. */
int khg_model_mle_update_sharded(khg_ctx *ctx, khg_model *m, khg_accs *a, const khg_mle_options *o, uint16_t flags, void *comm,
                                 int32_t nranks, int32_t rank, float *objf_change, float *count, int32_t *floored_elems,
                                 int32_t *floored_gauss, int32_t *removed);
/* Its pieces, for callers that move the rows themselves (ranks that cannot share device buffers): the update of pdfs [first_pdf,
 * first_pdf + n_pdf) alone (rows rewritten in place in the old layout, one 32-byte result per pdf kept on the handle); the rows of a
 * range + its results to / from the host (any pointer may be NULL); the finish on the complete rows and results. */
int khg_model_mle_update_range(khg_ctx *ctx, khg_model *m, const khg_accs *a, const khg_mle_options *o, uint16_t flags,
                               int32_t first_pdf, int32_t n_pdf);
int khg_model_mle_rows_download(khg_ctx *ctx, khg_model *m, int32_t first_pdf, int32_t n_pdf, float *weights_h, float *gconsts_h,
                                float *means_invvars_h, float *inv_vars_h, void *results_h /* 32 n_pdf bytes */);
int khg_model_mle_rows_upload(khg_ctx *ctx, khg_model *m, int32_t first_pdf, int32_t n_pdf, const float *weights_h,
                              const float *gconsts_h, const float *means_invvars_h, const float *inv_vars_h, const void *results_h);
int khg_model_mle_update_finish(khg_ctx *ctx, khg_model *m, float *objf_change, float *count, int32_t *floored_elems,
                                int32_t *floored_gauss, int32_t *removed);
/* Mixing up on the handle: AmDiagGmm::SplitByCount's per-pdf DiagGmm::Split (csrc/am-diag-gmm.cc:72-90, csrc/diag-gmm.cc:780-851)
 * to targets_h[p] >= current components (the caller computes them with GetSplitTargets, csrc/model-common.cc:29-70, from
 * the per-pdf occupancies -- khg_accs_download_range(0, sumG)).  The reference draws the perturbations from the process-global
 * rand(); here they are INJECTED: randn_h holds n_randn >= (sum of new components) x dim standard normal deviates, consumed pdf
 * by pdf in split order (KHG_E_ARG when there are fewer), so every rank of a multi-GPU job (and the host form) perturbs identically.  Parameters bit-identical to
 * the host form, gconsts through logf.  The handle is updated in place (call khg_accs_relayout afterwards). */
int khg_model_split(khg_ctx *ctx, khg_model *m, const int32_t *targets_h, float perturb_factor, const float *randn_h,
                    int64_t n_randn);
/* Mixing down on the handle: AmDiagGmm::MergeByCount's per-pdf DiagGmm::Merge (csrc/am-diag-gmm.cc:91-108, csrc/diag-gmm.cc:557-759,
 * MergedComponentsLogdet :761-778) to 1 <= targets_h[p] <= current components (GetSplitTargets again, with "can't merge below 1"
 * applied by the caller).  One workgroup per pdf, float arithmetic in the order of khg_diag_gmm_merge; the pair merged at each
 * step is the first maximum of the lower triangle in (i, j < i) order, whatever the thread count.  The handle is updated in
 * place (call khg_accs_relayout afterwards). */
int khg_model_merge(khg_ctx *ctx, khg_model *m, const int32_t *targets_h);
/* total Gaussians and (gauss_off_h may be NULL) the current gauss_off[num_pdfs+1] of the handle */
int khg_model_num_gauss(const khg_model *m, int64_t *total, int32_t *gauss_off_h);
/* parameters back to the host (AmDiagGmm::Write needs them); any pointer may be NULL */
int khg_model_download(khg_ctx *ctx, const khg_model *m, float *weights_h, float *gconsts_h,
                       float *means_invvars_h, float *inv_vars_h);
/* re-lay the accumulator block for the handle's current gauss_off and zero it */
int khg_accs_relayout(khg_ctx *ctx, khg_accs *a, const khg_model *m);
/* only the transition statistics [num_tids+1] and the 8 scalars (what TransitionModel::MleUpdate and the
 * log line of scripts/gmm_acc_stats_ali.py need) -- a few kB instead of the whole block */
int khg_accs_download_trans(khg_ctx *ctx, const khg_accs *a, double *trans_h, double *scalars_h);

/* any slice [first, first+count) of the fp64 block (e.g. occ = [0, sumG) for the mix-up targets of
 * scripts/gmm_est.py:66-70) */
int khg_accs_download_range(khg_ctx *ctx, const khg_accs *a, int64_t first, int64_t count,
                            double *dst_h);
/* scripts/gmm_boost_silence.py:10-45 on the handle: weights of the listed pdfs *= scale, their gconsts
 * recomputed (DiagGmm::SetWeights + ComputeGconsts, csrc/diag-gmm.cc:103-147), tile image repacked */
int khg_model_scale_weights(khg_ctx *ctx, khg_model *m, int32_t n, const int32_t *pdfs_h,
                            float scale);

/* TransitionModel::MleUpdate (csrc/transition-model.cc:657-750) + ComputeDerivedOfProbs (:339-359) */
int khg_transition_mle_update(int32_t num_tstates, const int32_t *state2id,
                              const int32_t *self_loop_of, const double *stats, float floor_,
                              float mincount, float *log_probs, float *non_self_loop_log_probs,
                              float *objf_impr, float *count);

/* GetScaledTransitionLogProb (csrc/hmm-utils.cc:442-463) negated, for every tid:
 * out_cost[0..num_tids] (entry 0 = 0). */
int khg_scaled_trans_cost(int32_t num_tids, const float *log_probs,
                          const float *non_self_loop_log_probs, const int32_t *id2state,
                          const uint8_t *is_self_loop, float transition_scale,
                          float self_loop_scale, float *out_cost);

#ifdef __cplusplus
}
#endif
#endif /* KHG_HIP_H_ */
