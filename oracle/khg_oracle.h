/* oracle/khg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's HMM-GMM EM hot path
 * (csukuangfj/kaldi-hmm-gmm v1.1.4). Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (kaldi_hmm_gmm_amd + libkhg_hip.so) never does.
 *
 * Pinning status: the GMM arithmetic (gconsts, log-likes, posteriors,
 * accumulators, flags) is pinned by the reference's own formula tests
 * (kaldi-hmm-gmm/python/tests/test_diag_gmm.py, test_mle_diag_gmm.py) and the
 * transition-model golden table (test_transition_model.py), all re-evaluated in
 * tests/test_oracle_pins.py.  The reference holds NO known-answer test for
 * FasterDecoder / AlignUtteranceWrapper / MleDiagGmmUpdate / AddTransitionProbs
 * and cannot be built here (Eigen, OpenFst, kaldifst, kaldi_native_io are
 * network-fetched): for those functions this oracle is "PARITY UNPINNED" -- a
 * line-by-line restatement cross-checked only by an independent exact Viterbi.
 * Exception (round 4): the decoder's HashList.  csrc/hash-list.h needs the standard
 * library only; oracle/ref_hashlist_harness.cc compiles it where it lies
 * (make -C oracle ref -> oracle/_ref/hashlist_ref) and tests/golden/hashlist_ref.json
 * holds its recorded answers, which the hl_* code below reproduces
 * (tests/test_oracle_pins.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/kaldi-hmm-gmm/csrc/).
 */
#ifndef KHG_ORACLE_H_
#define KHG_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define ORC_OK 0
#define ORC_ERR_NAN (-1)        /* KHG_ERR "not a number in gconst" / "Invalid answer" */
#define ORC_ERR_ARG (-2)        /* KHG_ASSERT / bad beams / bad labels */
#define ORC_ERR_NOMEM (-3)

/* ---- model: ragged AmDiagGmm as flat arrays (am-diag-gmm.h:96, diag-gmm.h:243-256) ---- */
typedef struct {
  int32_t num_pdfs;
  int32_t dim;
  const int32_t *gauss_off;   /* [num_pdfs+1] */
  const float *gconsts;       /* [sumG] */
  const float *means_invvars; /* [sumG*dim] row-major */
  const float *inv_vars;      /* [sumG*dim] row-major */
} orc_model;

/* ---- decoding graph: StdVectorFst as CSR by source state ---- */
typedef struct {
  int32_t num_states;
  int32_t start;             /* -1 == kNoStateId */
  const int32_t *arc_off;    /* [num_states+1] */
  const int32_t *ilabel;     /* [num_arcs] */
  const int32_t *olabel;
  const float *weight;       /* tropical */
  const int32_t *nextstate;
  const float *final;        /* [num_states]; +inf == Weight::Zero() (not final) */
} orc_graph;

/* diag-gmm.cc:103-147 DiagGmm::ComputeGconsts. returns ORC_OK / ORC_ERR_NAN */
int orc_compute_gconsts(int32_t G, int32_t D, const float *weights, const float *inv_vars,
                        const float *means_invvars, float *gconsts, int32_t *num_bad);

/* diag-gmm.cc:167-176 DiagGmm::LogLikelihoods (one frame, one pdf) */
void orc_loglikes(int32_t G, int32_t D, const float *gconsts, const float *means_invvars,
                  const float *inv_vars, const float *x, float *out);

/* Same expression evaluated in the order the HIP MFMA kernel uses:
 * s = gconst; for d in steps of 2: fmaf(miv[d]x[d]), fmaf(miv[d+1]x[d+1]), fmaf(-0.5iv[d]x[d]^2), fmaf(-0.5iv[d+1]x[d+1]^2).
 * Not a reference function: lets tests separate "contraction order" from "kernel bug". */
void orc_loglikes_fma_order(int32_t G, int32_t D, const float *gconsts, const float *means_invvars,
                            const float *inv_vars, const float *x, float *out);

/* eigen.cc:14-18 */
float orc_logsumexp(int32_t n, const float *v);
/* eigen.cc:20-32; returns log-sum-exp, writes posteriors */
float orc_softmax(int32_t n, const float *v, float *out);

/* diag-gmm.cc:150-165 DiagGmm::LogLikelihood: ORC_ERR_NAN when NaN/Inf */
int orc_gmm_loglike(const orc_model *m, int32_t pdf, const float *x, float *out);
/* diag-gmm.cc:368-392 DiagGmm::ComponentPosteriors */
int orc_component_posteriors(const orc_model *m, int32_t pdf, const float *x, float *post,
                             float *log_like);

/* decodable-am-diag-gmm.cc:29-71 for every (frame, pdf in list): out[j*T + t] */
int orc_loglikes_matrix(const orc_model *m, int32_t T, const float *feats, int32_t npdf,
                        const int32_t *pdfs, float *out);

/* hmm-utils.cc:442-493 AddTransitionProbs on CSR arcs (weights modified in place).
 * log_probs[num_tids+1], non_self_loop_log_probs[num_tstates+1], id2state[num_tids+1],
 * is_self_loop[num_tids+1]. */
int orc_add_transition_probs(int32_t num_arcs, const int32_t *ilabel, float *weight,
                             int32_t num_tids, const float *log_probs,
                             const float *non_self_loop_log_probs, const int32_t *id2state,
                             const uint8_t *is_self_loop, float transition_scale,
                             float self_loop_scale, int32_t num_disambig,
                             const int32_t *disambig_sorted);

/* decoder-wrappers.cc:111-140 ModifyGraphForCarefulAlignment (+ OpenFst Concat semantics).
 * Output arrays must be sized: states 2*S+1, arcs 2*A+1+nfinal(<=S). Returns new sizes. */
int orc_careful_graph(const orc_graph *g, int32_t *out_num_states, int32_t *out_start,
                      int32_t *arc_off, int32_t *ilabel, int32_t *olabel, float *weight,
                      int32_t *nextstate, float *final, int32_t *out_num_arcs);

typedef struct {
  float beam;          /* decoder-wrappers.h:23-37 AlignConfig */
  float retry_beam;
  int32_t careful;     /* caller applies orc_careful_graph first; kept for bookkeeping */
  /* faster-decoder.h:24-49 (AlignUtteranceWrapper only overrides beam) */
  int32_t max_active;
  int32_t min_active;
  float beam_delta;
  float hash_ratio;
} orc_align_config;

void orc_align_config_default(orc_align_config *c);

/* status values written by orc_align_utterance */
#define ORC_ALIGN_DONE 0
#define ORC_ALIGN_ERROR 1      /* empty graph / no final state reached / no best path */
#define ORC_ALIGN_RETRIED 2    /* bit flag OR-ed in when the retry beam was used */

typedef struct {
  int64_t loglike_evals;   /* decodable cache misses (for the CPU baseline's FLOP count) */
  int64_t tokens_expanded;
} orc_align_stats;

/* decoder-wrappers.cc:16-108 AlignUtteranceWrapper + faster-decoder.cc (whole) +
 * decodable-am-diag-gmm.{h,cc} DecodableAmDiagGmmScaled, line-faithful (HashList order,
 * float tmp_array_ in GetCutoff, double token costs).
 *   alignment[T], words[<=max_words] outputs; *num_words; *like (float, :95).
 * returns ORC_OK, or ORC_ERR_ARG (bad beams -> KHG_ERR :29-33), ORC_ERR_NAN (decodable :63-65). */
int orc_align_utterance(const orc_align_config *cfg, float acoustic_scale, const orc_graph *g,
                        const orc_model *m, const int32_t *id2pdf /*[num_tids+1]*/,
                        int32_t num_tids, int32_t T, const float *feats, int32_t *alignment,
                        int32_t *words, int32_t max_words, int32_t *num_words, float *like,
                        int32_t *status, orc_align_stats *stats);

/* Same decoder but reading acoustic log-likes from a matrix ll[j*ll_stride + t] for the pdf
 * list `pdfs` (sorted) instead of evaluating the GMM: used to check the HIP Viterbi kernel
 * bit-exactly on identical scores. */
int orc_align_utterance_ll(const orc_align_config *cfg, float acoustic_scale, const orc_graph *g,
                           const int32_t *id2pdf, int32_t num_tids, int32_t T, int32_t npdf,
                           const int32_t *pdfs, const float *ll, int64_t ll_stride,
                           int32_t *alignment, int32_t *words, int32_t max_words,
                           int32_t *num_words, float *like, int32_t *status,
                           orc_align_stats *stats);

/* Independent second opinion: exact (unpruned) Viterbi in the same arithmetic as the token
 * costs (faster-decoder.h:119-137): double path cost, float arc weight, float ac_cost.
 * best_cost includes the final weight. status as above (ERROR when no final reachable). */
int orc_exact_viterbi_ll(float acoustic_scale, const orc_graph *g, const int32_t *id2pdf,
                         int32_t num_tids, int32_t T, int32_t npdf, const int32_t *pdfs,
                         const float *ll, int64_t ll_stride, int32_t *alignment,
                         double *best_cost, int32_t *status);

/* ---- accumulation: mle-am-diag-gmm.cc:41-52, mle-diag-gmm.cc:123-158,
 *      transition-model.h:183-189 as driven by scripts/gmm_acc_stats_ali.py:46-56 ---- */
typedef struct {
  double *occ;         /* [sumG] */
  double *mean_acc;    /* [sumG*dim] */
  double *var_acc;     /* [sumG*dim] */
  double *trans_acc;   /* [num_tids+1] */
  double total_frames;
  double total_log_like;
} orc_accs;

/* returns sum of per-frame log-likes as the python loop does (double += float) in *log_like */
int orc_acc_stats_ali(const orc_model *m, const int32_t *id2pdf, int32_t num_tids, int32_t T,
                      const float *feats, const int32_t *ali, float weight, orc_accs *accs,
                      double *log_like);

/* ---- M-step: mle-diag-gmm.cc:243-390,479-499; diag-gmm-normal.cc:14-48;
 *      diag-gmm.cc:853-938 (RemoveComponents) ---- */
typedef struct {
  float min_gaussian_weight;      /* 1e-5 */
  float min_gaussian_occupancy;   /* 10 */
  double min_variance;            /* 1e-3 */
  int32_t remove_low_count_gaussians; /* 1 */
  const double *variance_floor_vector; /* mle-diag-gmm.h:26-28: per-dimension floor, NULL = not supplied */
} orc_mle_opts;
void orc_mle_opts_default(orc_mle_opts *o);

/* flags: model-common.h:18-26 m=1 v=2 w=4 t=8 */
uint16_t orc_augment_gmm_flags(uint16_t flags); /* model-common.cc:72-85 */

/* One pdf. Arrays are updated in place; *G may shrink (removed Gaussians compacted).
 * acc_flags = flags the accumulator was created with (already augmented). */
int orc_mle_diag_gmm_update(const orc_mle_opts *o, int32_t *G, int32_t D, const double *occ,
                            const double *mean_acc, const double *var_acc, uint16_t acc_flags,
                            uint16_t flags, float *weights, float *gconsts, float *means_invvars,
                            float *inv_vars, float *obj_change, float *count,
                            int32_t *floored_elems, int32_t *floored_gauss, int32_t *removed);

/* diag-gmm.cc:557-759 DiagGmm::Merge (+ MergedComponentsLogdet :761-778): arrays updated in place, *G shrinks to
 * target_components; history[2 * (G_in - target)] receives the merged pairs (max_i, max_j) in order (NULL allowed).
 * Sums run over d (and over components for the global mean / variance) in index order, float arithmetic as written. */
int orc_diag_gmm_merge(int32_t *G, int32_t D, int32_t target_components, float *weights, float *gconsts,
                       float *means_invvars, float *inv_vars, int32_t *history, int32_t *num_history);

/* mle-diag-gmm.cc:479-499 */
float orc_ml_objective(int32_t G, int32_t D, const float *gconsts, const float *means_invvars,
                       const float *inv_vars, const double *occ, const double *mean_acc,
                       const double *var_acc, uint16_t acc_flags);

/* transition-model.cc:657-750 TransitionModel::MleUpdate (share_for_pdfs=false) +
 * ComputeDerivedOfProbs (:339-359). state2id[num_tstates+2], self_loop_of[num_tstates+1]. */
int orc_transition_mle_update(int32_t num_tstates, const int32_t *state2id,
                              const int32_t *self_loop_of, const double *stats, float floor_,
                              float mincount, float *log_probs, float *non_self_loop_log_probs,
                              float *objf_impr, float *count);

/* ---- test hooks: the decoder's HashList restatement (hash-list.h, hash-list-inl.h) driven the way the reference's
 * csrc/hash-list-test.cc drives HashList<Int, T> ---- */
void *orc_hl_create(void);
void orc_hl_destroy(void *h);
void orc_hl_set_size(void *h, int64_t size);
int orc_hl_find(void *h, int32_t key, int64_t *val);
void orc_hl_put(void *h, int32_t key, int64_t val);
int orc_hl_insert(void *h, int32_t key, int64_t val);
int64_t orc_hl_list(void *h, int32_t *keys, int64_t *vals, int64_t cap);
int64_t orc_hl_clear_reinsert(void *h, int64_t new_size, int32_t shift);
void orc_hl_drop(void *h);

/* bench.py's cpu_baseline, variant B: `num_threads` POSIX threads run orc_align_utterance +
 * orc_acc_stats_ali per utterance (private accumulators) over utterances [first_utt, first_utt + n_utt) of a set in
 * the C-ABI's concatenated CSR layout, for at most `budget_seconds`.  Not a reference function (the reference is
 * single-threaded). */
int orc_em_pass_mt(const orc_align_config *cfg, float acoustic_scale, const orc_model *m,
                   const int32_t *id2pdf, int32_t num_tids, int32_t first_utt, int32_t n_utt,
                   const int64_t *frame_off, const float *feats, const int64_t *state_off,
                   const int32_t *start, const int64_t *arc_off, const int32_t *ilabel,
                   const int32_t *olabel, const float *weight, const int32_t *nextstate,
                   const float *final, int32_t num_threads, double budget_seconds,
                   int64_t *frames_done, int32_t *utts_done, int32_t *failed, double *seconds);
/* the same pass, keeping what it computed (any of the four may be NULL): alignments (0 on failed utterances; positioned by
 * frame_off relative to first_utt), per-utterance status (-1 = not reached within the budget; the reached utterances are a
 * prefix) and like, and the sum of the threads' accumulators (caller-allocated, zeroed) -- the oracle's answer for a parity
 * check at the benchmark's own scale */
int orc_em_pass_mt_keep(const orc_align_config *cfg, float acoustic_scale, const orc_model *m,
                        const int32_t *id2pdf, int32_t num_tids, int32_t first_utt, int32_t n_utt,
                        const int64_t *frame_off, const float *feats, const int64_t *state_off,
                        const int32_t *start, const int64_t *arc_off, const int32_t *ilabel,
                        const int32_t *olabel, const float *weight, const int32_t *nextstate,
                        const float *final, int32_t num_threads, double budget_seconds,
                        int64_t *frames_done, int32_t *utts_done, int32_t *failed, double *seconds,
                        int32_t *ali_out, int32_t *status_out, float *like_out, orc_accs *acc_sum);

#ifdef __cplusplus
}
#endif
#endif /* KHG_ORACLE_H_ */
