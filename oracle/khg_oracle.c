/* oracle/khg_oracle.c -- TEST INFRASTRUCTURE ONLY (see khg_oracle.h header).
 *
 * Plain-C CPU restatement of the reference's EM hot path.  Paths cited are relative to
 * /root/reference/kaldi-hmm-gmm/csrc/.  Eigen's internal summation order for the two
 * G x D gemv products and its vectorised exp are not reproducible without Eigen, so the
 * dot products here are sequential float sums and exp/log are libm's: the reference's own
 * tests accept 1e-4 on these quantities (python/tests/test_diag_gmm.py:342-349).
 *
 * FasterDecoder / AlignUtteranceWrapper / M-step: "PARITY UNPINNED" (no known-answer test in
 * the reference; reference not buildable offline) -- line-faithful restatement only; the
 * decoder's HashList alone is pinned by the reference's own header (oracle/_ref, see khg_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math: float/double rounding points matter).
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime / pthreads for orc_em_pass_mt under -std=c11 */
#include "khg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_LOG_2PI
#define M_LOG_2PI 1.8378770664093454835606594728112 /* kaldi-math.h */
#endif

/* ------------------------------------------------------------------------- */
/* diag-gmm.cc:103-147                                                       */
int orc_compute_gconsts(int32_t G, int32_t D, const float *weights, const float *inv_vars,
                        const float *means_invvars, float *gconsts, int32_t *num_bad_out) {
  float offset = -0.5 * M_LOG_2PI * D; /* :106 double expression stored to float */
  int32_t num_bad = 0;
  for (int32_t mix = 0; mix < G; ++mix) {
    if (!(weights[mix] >= 0)) return ORC_ERR_ARG; /* :114 KHG_ASSERT */
    float gc = logf(weights[mix]) + offset;       /* :117 */
    for (int32_t d = 0; d < D; ++d) {
      float iv = inv_vars[(size_t)mix * D + d], mi = means_invvars[(size_t)mix * D + d];
      /* :121-123: 0.5 is a double literal, so the right-hand side is evaluated in double and
       * the += rounds back to float on every d. */
      gc += 0.5 * logf(iv) - 0.5 * mi * mi / iv;
    }
    if (isnan(gc)) return ORC_ERR_NAN; /* :131-134 */
    if (isinf(gc)) {                   /* :135-140 */
      num_bad++;
      if (gc > 0) gc = -gc;
    }
    gconsts[mix] = gc;
  }
  if (num_bad_out) *num_bad_out = num_bad;
  return ORC_OK;
}

/* diag-gmm.cc:167-176: gconsts + means_invvars*x - 0.5*inv_vars*x.^2 (all float) */
void orc_loglikes(int32_t G, int32_t D, const float *gconsts, const float *means_invvars,
                  const float *inv_vars, const float *x, float *out) {
  for (int32_t g = 0; g < G; ++g) {
    const float *mi = means_invvars + (size_t)g * D, *iv = inv_vars + (size_t)g * D;
    float a = 0.0f, b = 0.0f;
    for (int32_t d = 0; d < D; ++d) a += mi[d] * x[d];
    for (int32_t d = 0; d < D; ++d) {
      float xx = x[d] * x[d]; /* data.array().square() */
      b += iv[d] * xx;
    }
    out[g] = (gconsts[g] + a) - 0.5f * b;
  }
}

void orc_loglikes_fma_order(int32_t G, int32_t D, const float *gconsts,
                            const float *means_invvars, const float *inv_vars, const float *x,
                            float *out) {
  /* K1's MFMA step s consumes (M[2s]x[2s], M[2s+1]x[2s+1], V'[2s]x2[2s], V'[2s+1]x2[2s+1]), V' = -0.5 V */
  for (int32_t g = 0; g < G; ++g) {
    const float *mi = means_invvars + (size_t)g * D, *iv = inv_vars + (size_t)g * D;
    float s = gconsts[g];
    for (int32_t d = 0; d < D; d += 2) {
      int has2 = d + 1 < D;
      s = fmaf(mi[d], x[d], s);
      if (has2) s = fmaf(mi[d + 1], x[d + 1], s);
      s = fmaf(-0.5f * iv[d], x[d] * x[d], s);
      if (has2) s = fmaf(-0.5f * iv[d + 1], x[d + 1] * x[d + 1], s);
    }
    out[g] = s;
  }
}

/* eigen.cc:14-18 */
float orc_logsumexp(int32_t n, const float *v) {
  float max_v = v[0];
  for (int32_t i = 1; i < n; ++i)
    if (v[i] > max_v) max_v = v[i];
  float s = 0.0f;
  for (int32_t i = 0; i < n; ++i) s += expf(v[i] - max_v);
  return logf(s) + max_v;
}

/* eigen.cc:20-32 */
float orc_softmax(int32_t n, const float *v, float *out) {
  float max_v = v[0];
  for (int32_t i = 1; i < n; ++i)
    if (v[i] > max_v) max_v = v[i];
  float s = 0.0f;
  for (int32_t i = 0; i < n; ++i) {
    out[i] = expf(v[i] - max_v);
    s += out[i];
  }
  float lse = logf(s) + max_v;
  for (int32_t i = 0; i < n; ++i) out[i] = out[i] / s;
  return lse;
}

static int pdf_loglikes_tmp(const orc_model *m, int32_t pdf, const float *x, float **buf,
                            int32_t *G) {
  int32_t g0 = m->gauss_off[pdf], g1 = m->gauss_off[pdf + 1];
  *G = g1 - g0;
  *buf = (float *)malloc(sizeof(float) * (size_t)(*G > 0 ? *G : 1));
  if (!*buf) return ORC_ERR_NOMEM;
  orc_loglikes(*G, m->dim, m->gconsts + g0, m->means_invvars + (size_t)g0 * m->dim,
               m->inv_vars + (size_t)g0 * m->dim, x, *buf);
  return ORC_OK;
}

/* diag-gmm.cc:150-165 */
int orc_gmm_loglike(const orc_model *m, int32_t pdf, const float *x, float *out) {
  float *ll;
  int32_t G;
  int rc = pdf_loglikes_tmp(m, pdf, x, &ll, &G);
  if (rc) return rc;
  float log_sum = orc_logsumexp(G, ll);
  free(ll);
  *out = log_sum;
  if (isnan(log_sum) || isinf(log_sum)) return ORC_ERR_NAN; /* :160-162 */
  return ORC_OK;
}

/* diag-gmm.cc:368-392 */
int orc_component_posteriors(const orc_model *m, int32_t pdf, const float *x, float *post,
                             float *log_like) {
  float *ll;
  int32_t G;
  int rc = pdf_loglikes_tmp(m, pdf, x, &ll, &G);
  if (rc) return rc;
  float log_sum = orc_softmax(G, ll, post);
  free(ll);
  *log_like = log_sum;
  if (isnan(log_sum) || isinf(log_sum)) return ORC_ERR_NAN; /* :385-387 */
  return ORC_OK;
}

/* decodable-am-diag-gmm.cc:55-65 for a (frame x pdf-list) block */
int orc_loglikes_matrix(const orc_model *m, int32_t T, const float *feats, int32_t npdf,
                        const int32_t *pdfs, float *out) {
  int32_t maxG = 1;
  for (int32_t j = 0; j < npdf; ++j) {
    int32_t G = m->gauss_off[pdfs[j] + 1] - m->gauss_off[pdfs[j]];
    if (G > maxG) maxG = G;
  }
  float *ll = (float *)malloc(sizeof(float) * (size_t)maxG);
  if (!ll) return ORC_ERR_NOMEM;
  int rc = ORC_OK;
  for (int32_t j = 0; j < npdf; ++j) {
    int32_t p = pdfs[j], g0 = m->gauss_off[p], G = m->gauss_off[p + 1] - g0;
    for (int32_t t = 0; t < T; ++t) {
      orc_loglikes(G, m->dim, m->gconsts + g0, m->means_invvars + (size_t)g0 * m->dim,
                   m->inv_vars + (size_t)g0 * m->dim, feats + (size_t)t * m->dim, ll);
      float v = orc_logsumexp(G, ll);
      if (isnan(v) || isinf(v)) rc = ORC_ERR_NAN;
      out[(size_t)j * T + t] = v;
    }
  }
  free(ll);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* hmm-utils.cc:442-463 GetScaledTransitionLogProb                           */
static float scaled_transition_log_prob(int32_t tid, const float *log_probs,
                                        const float *nsl_log_probs, const int32_t *id2state,
                                        const uint8_t *is_self_loop, float transition_scale,
                                        float self_loop_scale) {
  if (transition_scale == self_loop_scale) {
    return log_probs[tid] * transition_scale;
  } else {
    if (is_self_loop[tid]) {
      return self_loop_scale * log_probs[tid];
    } else {
      int32_t ts = id2state[tid];
      /* transition-model.cc:514-520 GetTransitionLogProbIgnoringSelfLoops */
      float ignoring = log_probs[tid] - nsl_log_probs[ts];
      return self_loop_scale * nsl_log_probs[ts] + transition_scale * ignoring;
    }
  }
}

/* hmm-utils.cc:465-493 */
int orc_add_transition_probs(int32_t num_arcs, const int32_t *ilabel, float *weight,
                             int32_t num_tids, const float *log_probs,
                             const float *nsl_log_probs, const int32_t *id2state,
                             const uint8_t *is_self_loop, float transition_scale,
                             float self_loop_scale, int32_t num_disambig,
                             const int32_t *disambig_sorted) {
  for (int32_t a = 0; a < num_arcs; ++a) {
    int32_t l = ilabel[a];
    if (l >= 1 && l <= num_tids) {
      float s = scaled_transition_log_prob(l, log_probs, nsl_log_probs, id2state, is_self_loop,
                                           transition_scale, self_loop_scale);
      /* Times(arc.weight, TropicalWeight(-s)): float add; Zero (+inf) stays +inf */
      weight[a] = weight[a] + (-s);
    } else if (l != 0) {
      int found = 0;
      for (int32_t i = 0; i < num_disambig; ++i)
        if (disambig_sorted[i] == l) found = 1;
      if (!found) return ORC_ERR_ARG; /* :484-488 KHG_ERR */
    }
  }
  return ORC_OK;
}

/* decoder-wrappers.cc:111-140 + OpenFst Concat(MutableFst*, const Fst&):
 * rhs = copy of fst with all finals removed, plus a new pre-initial state (index S in rhs)
 * that is final with weight One and has an eps arc to rhs's old start; rhs states are
 * appended after fst's (offset S); every final state f of fst loses its final weight and
 * gets an eps arc (0,0,final(f)) to rhs.start + S.  The new arc is appended after f's
 * existing arcs (AddArc). */
int orc_careful_graph(const orc_graph *g, int32_t *out_num_states, int32_t *out_start,
                      int32_t *arc_off, int32_t *ilabel, int32_t *olabel, float *weight,
                      int32_t *nextstate, float *final, int32_t *out_num_arcs) {
  int32_t S = g->num_states;
  if (S == 0) return ORC_ERR_ARG;
  int32_t na = 0;
  int32_t rhs_start = S + S; /* pre_initial index inside the concatenation */
  for (int32_t s = 0; s < S; ++s) {
    arc_off[s] = na;
    for (int32_t a = g->arc_off[s]; a < g->arc_off[s + 1]; ++a) {
      ilabel[na] = g->ilabel[a]; olabel[na] = g->olabel[a]; weight[na] = g->weight[a];
      nextstate[na] = g->nextstate[a]; ++na;
    }
    if (g->final[s] != INFINITY) {
      ilabel[na] = 0; olabel[na] = 0; weight[na] = g->final[s]; nextstate[na] = rhs_start; ++na;
    }
    final[s] = INFINITY;
  }
  for (int32_t s = 0; s < S; ++s) {
    arc_off[S + s] = na;
    for (int32_t a = g->arc_off[s]; a < g->arc_off[s + 1]; ++a) {
      ilabel[na] = g->ilabel[a]; olabel[na] = g->olabel[a]; weight[na] = g->weight[a];
      nextstate[na] = g->nextstate[a] + S; ++na;
    }
    final[S + s] = INFINITY;
  }
  arc_off[2 * S] = na;
  ilabel[na] = 0; olabel[na] = 0; weight[na] = 0.0f; nextstate[na] = g->start + S; ++na;
  final[2 * S] = 0.0f;
  arc_off[2 * S + 1] = na;
  *out_num_states = 2 * S + 1;
  *out_start = g->start;
  *out_num_arcs = na;
  return ORC_OK;
}

void orc_align_config_default(orc_align_config *c) {
  c->beam = 200.0f; c->retry_beam = 0.0f; c->careful = 0;
  c->max_active = INT32_MAX; c->min_active = 20; c->beam_delta = 0.5f; c->hash_ratio = 2.0f;
}

/* ------------------------------------------------------------------------- */
/* Decodable: decodable-am-diag-gmm.{h,cc}                                    */
typedef struct {
  /* GMM-evaluating flavour */
  const orc_model *m;
  const float *feats;
  float *cache_ll;      /* [num_pdfs] */
  int32_t *cache_time;  /* [num_pdfs] */
  float *tmp;           /* [maxG] */
  /* matrix flavour */
  const float *ll;
  int64_t ll_stride;
  int32_t *pdf2col;     /* [max_pdf+1], -1 when absent */
  int32_t pdf2col_n;
  /* common */
  const int32_t *id2pdf;
  int32_t num_tids;
  int32_t T;
  float scale;
  int err;
  orc_align_stats *stats;
} Decodable;

/* decodable-am-diag-gmm.h:95-98 + .cc:29-71 */
static float dec_loglike(Decodable *d, int32_t frame, int32_t tid) {
  if (tid < 1 || tid > d->num_tids) { d->err = ORC_ERR_ARG; return 0.0f; }
  int32_t pdf = d->id2pdf[tid];
  float v;
  if (d->m) {
    if (pdf < 0 || pdf >= d->m->num_pdfs) { d->err = ORC_ERR_ARG; return 0.0f; }
    if (d->cache_time[pdf] == frame) {
      v = d->cache_ll[pdf]; /* .cc:37-39 */
    } else {
      int32_t g0 = d->m->gauss_off[pdf], G = d->m->gauss_off[pdf + 1] - g0, D = d->m->dim;
      orc_loglikes(G, D, d->m->gconsts + g0, d->m->means_invvars + (size_t)g0 * D,
                   d->m->inv_vars + (size_t)g0 * D, d->feats + (size_t)frame * D, d->tmp);
      v = orc_logsumexp(G, d->tmp);                     /* .cc:61 */
      if (isnan(v) || isinf(v)) d->err = ORC_ERR_NAN;   /* .cc:63-65 */
      d->cache_ll[pdf] = v;
      d->cache_time[pdf] = frame;
      if (d->stats) d->stats->loglike_evals++;
    }
  } else {
    if (pdf < 0 || pdf >= d->pdf2col_n || d->pdf2col[pdf] < 0) { d->err = ORC_ERR_ARG; return 0.0f; }
    v = d->ll[(int64_t)d->pdf2col[pdf] * d->ll_stride + frame];
    if (isnan(v) || isinf(v)) d->err = ORC_ERR_NAN;
  }
  return d->scale * v; /* .h:96 float * float */
}

/* ------------------------------------------------------------------------- */
/* FasterDecoder: faster-decoder.{h,cc}, hash-list{.h,-inl.h}                 */
typedef struct Token {
  int32_t ilabel, olabel, nextstate;
  float weight;
  struct Token *prev;
  double cost; /* faster-decoder.h:119 */
} Token;

typedef struct Elem {
  int32_t key;
  Token *val;
  struct Elem *tail;
} Elem;

typedef struct {
  size_t prev_bucket;
  Elem *last_elem;
} Bucket;

#define NOBUCKET ((size_t)-1)

typedef struct PoolBlock {
  struct PoolBlock *next;
  size_t used, cap;
  /* payload follows */
} PoolBlock;

typedef struct {
  PoolBlock *head;
  size_t item;
} Pool;

static void *pool_alloc(Pool *p) {
  if (!p->head || p->head->used == p->head->cap) {
    size_t cap = 4096;
    PoolBlock *b = (PoolBlock *)malloc(sizeof(PoolBlock) + cap * p->item);
    if (!b) return NULL;
    b->next = p->head; b->used = 0; b->cap = cap; p->head = b;
  }
  return (char *)(p->head + 1) + (p->head->used++) * p->item;
}
static void pool_free_all(Pool *p) {
  while (p->head) { PoolBlock *n = p->head->next; free(p->head); p->head = n; }
}

typedef struct {
  /* HashList (hash-list.h:104-127) */
  Elem *list_head;
  size_t bucket_list_tail;
  size_t hash_size;
  Bucket *buckets;
  size_t nbuckets;
  Elem *freed_head;
  Pool elem_pool, tok_pool;
  /* FasterDecoder */
  const orc_graph *g;
  orc_align_config cfg;
  int32_t num_frames_decoded;
  const Elem **queue; size_t qn, qcap;
  float *tmp_array; size_t tn, tcap; /* std::vector<float> tmp_array_ (faster-decoder.h:188) */
  int oom;
  orc_align_stats *stats;
} Decoder;

/* hash-list-inl.h:28-37 SetSize */
static void hl_set_size(Decoder *d, size_t size) {
  d->hash_size = size;
  if (size > d->nbuckets) {
    Bucket *nb = (Bucket *)realloc(d->buckets, size * sizeof(Bucket));
    if (!nb) { d->oom = 1; return; }
    for (size_t i = d->nbuckets; i < size; ++i) { nb[i].prev_bucket = 0; nb[i].last_elem = NULL; }
    d->buckets = nb; d->nbuckets = size;
  }
}
/* hash-list-inl.h:39-54 Clear */
static Elem *hl_clear(Decoder *d) {
  for (size_t b = d->bucket_list_tail; b != NOBUCKET; b = d->buckets[b].prev_bucket)
    d->buckets[b].last_elem = NULL;
  d->bucket_list_tail = NOBUCKET;
  Elem *ans = d->list_head;
  d->list_head = NULL;
  return ans;
}
/* hash-list-inl.h:61-65 Delete */
static void hl_delete(Decoder *d, Elem *e) { e->tail = d->freed_head; d->freed_head = e; }
/* hash-list-inl.h:87-102 New */
static Elem *hl_new(Decoder *d) {
  if (d->freed_head) { Elem *a = d->freed_head; d->freed_head = a->tail; return a; }
  Elem *e = (Elem *)pool_alloc(&d->elem_pool);
  if (!e) d->oom = 1;
  return e;
}
/* hash-list-inl.h:129-174 Insert */
static Elem *hl_insert(Decoder *d, int32_t key, Token *val) {
  size_t index = (size_t)key % d->hash_size;
  Bucket *bucket = &d->buckets[index];
  if (bucket->last_elem != NULL) {
    Elem *head = (bucket->prev_bucket == NOBUCKET ? d->list_head
                                                  : d->buckets[bucket->prev_bucket].last_elem->tail),
         *tail = bucket->last_elem->tail;
    for (Elem *e = head; e != tail; e = e->tail)
      if (e->key == key) return e;
  }
  Elem *elem = hl_new(d);
  if (!elem) return NULL;
  elem->key = key; elem->val = val;
  if (bucket->last_elem == NULL) {
    if (d->bucket_list_tail == NOBUCKET) d->list_head = elem;
    else d->buckets[d->bucket_list_tail].last_elem->tail = elem;
    elem->tail = NULL;
    bucket->last_elem = elem;
    bucket->prev_bucket = d->bucket_list_tail;
    d->bucket_list_tail = index;
  } else {
    elem->tail = bucket->last_elem->tail;
    bucket->last_elem->tail = elem;
    bucket->last_elem = elem;
  }
  return elem;
}

/* faster-decoder.h:120-137 Token ctors (ref-counting is memory management only; a pool
 * replaces it, tokens live until the decoder is destroyed). */
static Token *tok_new(Decoder *d, int32_t il, int32_t ol, float w, int32_t ns, float ac_cost,
                      int has_ac, Token *prev) {
  Token *t = (Token *)pool_alloc(&d->tok_pool);
  if (!t) { d->oom = 1; return NULL; }
  t->ilabel = il; t->olabel = ol; t->weight = w; t->nextstate = ns; t->prev = prev;
  if (has_ac) {
    if (prev) t->cost = prev->cost + w + ac_cost; /* :125 double + float + float, left to right */
    else t->cost = w + ac_cost;                   /* float sum widened */
  } else {
    if (prev) t->cost = prev->cost + w;           /* :135 */
    else t->cost = w;
  }
  return t;
}

static void queue_push(Decoder *d, const Elem *e) {
  if (d->qn == d->qcap) {
    size_t nc = d->qcap ? d->qcap * 2 : 256;
    const Elem **nq = (const Elem **)realloc((void *)d->queue, nc * sizeof(*nq));
    if (!nq) { d->oom = 1; return; }
    d->queue = nq; d->qcap = nc;
  }
  d->queue[d->qn++] = e;
}

/* faster-decoder.cc:58-118 */
static void process_nonemitting(Decoder *d, double cutoff) {
  const orc_graph *g = d->g;
  for (const Elem *e = d->list_head; e != NULL; e = e->tail) queue_push(d, e);
  while (d->qn && !d->oom) {
    const Elem *e = d->queue[--d->qn];
    int32_t state = e->key;
    Token *tok = e->val;
    if (tok->cost > cutoff) continue;
    for (int32_t a = g->arc_off[state]; a < g->arc_off[state + 1]; ++a) {
      if (g->ilabel[a] != 0) continue;
      Token *new_tok = tok_new(d, 0, g->olabel[a], g->weight[a], g->nextstate[a], 0.0f, 0, tok);
      if (!new_tok) return;
      if (new_tok->cost > cutoff) continue; /* prune (:89-92) */
      Elem *e_found = hl_insert(d, g->nextstate[a], new_tok);
      if (!e_found) return;
      if (e_found->val == new_tok) { queue_push(d, e_found); continue; }
      /* :104 `*(e_found->val) < *new_tok` means e_found->cost > new_tok->cost (:141-143) */
      if (e_found->val->cost > new_tok->cost) {
        e_found->val = new_tok;
        queue_push(d, e_found);
      }
    }
  }
}

static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}
/* value std::nth_element leaves at position n (the n-th order statistic) of [first,last);
 * a full sort gives the same value and the same "everything before n is <= it" post-condition */
static float nth_value(float *first, size_t count, size_t n) {
  qsort(first, count, sizeof(float), cmp_float);
  return first[n];
}

/* faster-decoder.cc:243-335 */
static double get_cutoff(Decoder *d, Elem *list_head, size_t *tok_count, float *adaptive_beam,
                         Elem **best_elem) {
  double best_cost = INFINITY;
  size_t count = 0;
  const orc_align_config *c = &d->cfg;
  if (c->max_active == INT32_MAX && c->min_active == 0) {
    for (Elem *e = list_head; e != NULL; e = e->tail, ++count) {
      double w = e->val->cost;
      if (w < best_cost) { best_cost = w; if (best_elem) *best_elem = e; }
    }
    if (tok_count) *tok_count = count;
    if (adaptive_beam) *adaptive_beam = c->beam;
    return best_cost + c->beam;
  }
  d->tn = 0;
  for (Elem *e = list_head; e != NULL; e = e->tail, ++count) {
    double w = e->val->cost;
    if (d->tn == d->tcap) {
      size_t nc = d->tcap ? d->tcap * 2 : 256;
      float *nt = (float *)realloc(d->tmp_array, nc * sizeof(float));
      if (!nt) { d->oom = 1; return 0; }
      d->tmp_array = nt; d->tcap = nc;
    }
    d->tmp_array[d->tn++] = (float)w; /* vector<float>::push_back(double) */
    if (w < best_cost) { best_cost = w; if (best_elem) *best_elem = e; }
  }
  if (tok_count) *tok_count = count;
  double beam_cutoff = best_cost + c->beam;
  double min_active_cutoff = INFINITY, max_active_cutoff = INFINITY;
  size_t limit = d->tn;
  if (d->tn > (size_t)c->max_active) {
    max_active_cutoff = nth_value(d->tmp_array, d->tn, (size_t)c->max_active);
    limit = (size_t)c->max_active; /* second nth_element is restricted to [begin, begin+max_active) */
  }
  if (max_active_cutoff < beam_cutoff) {
    if (adaptive_beam) *adaptive_beam = max_active_cutoff - best_cost + c->beam_delta;
    return max_active_cutoff;
  }
  if (d->tn > (size_t)c->min_active) {
    if (c->min_active == 0) min_active_cutoff = best_cost;
    else min_active_cutoff = nth_value(d->tmp_array, limit, (size_t)c->min_active);
  }
  if (min_active_cutoff > beam_cutoff) {
    if (adaptive_beam) *adaptive_beam = min_active_cutoff - best_cost + c->beam_delta;
    return min_active_cutoff;
  } else {
    *adaptive_beam = c->beam;
    return beam_cutoff;
  }
}

/* faster-decoder.cc:154-240 */
static double process_emitting(Decoder *d, Decodable *dec) {
  const orc_graph *g = d->g;
  int32_t frame = d->num_frames_decoded;
  Elem *last_toks = hl_clear(d);
  size_t tok_cnt = 0;
  float adaptive_beam = 0;
  Elem *best_elem = NULL;
  double weight_cutoff = get_cutoff(d, last_toks, &tok_cnt, &adaptive_beam, &best_elem);
  /* :337-344 PossiblyResizeHash */
  size_t new_sz = (size_t)((float)tok_cnt * d->cfg.hash_ratio);
  if (new_sz > d->hash_size) hl_set_size(d, new_sz);

  double next_weight_cutoff = INFINITY;
  if (best_elem) { /* :175-188 */
    int32_t state = best_elem->key;
    Token *tok = best_elem->val;
    for (int32_t a = g->arc_off[state]; a < g->arc_off[state + 1]; ++a) {
      if (g->ilabel[a] != 0) {
        float ac_cost = -1 * dec_loglike(dec, frame, g->ilabel[a]);
        double new_weight = g->weight[a] + tok->cost + ac_cost;
        if (new_weight + adaptive_beam < next_weight_cutoff)
          next_weight_cutoff = new_weight + adaptive_beam;
      }
    }
  }
  for (Elem *e = last_toks, *e_tail; e != NULL && !d->oom; e = e_tail) { /* :195-236 */
    int32_t state = e->key;
    Token *tok = e->val;
    if (tok->cost < weight_cutoff) {
      if (d->stats) d->stats->tokens_expanded++;
      for (int32_t a = g->arc_off[state]; a < g->arc_off[state + 1]; ++a) {
        if (g->ilabel[a] != 0) {
          float ac_cost = -1 * dec_loglike(dec, frame, g->ilabel[a]);
          double new_weight = g->weight[a] + tok->cost + ac_cost;
          if (new_weight < next_weight_cutoff) {
            Token *new_tok = tok_new(d, g->ilabel[a], g->olabel[a], g->weight[a],
                                     g->nextstate[a], ac_cost, 1, tok);
            if (!new_tok) break;
            Elem *e_found = hl_insert(d, g->nextstate[a], new_tok);
            if (!e_found) break;
            if (new_weight + adaptive_beam < next_weight_cutoff)
              next_weight_cutoff = new_weight + adaptive_beam;
            if (e_found->val != new_tok) {
              if (e_found->val->cost > new_tok->cost) e_found->val = new_tok; /* :218-227 */
            }
          }
        }
      }
    }
    e_tail = e->tail;
    hl_delete(d, e);
  }
  d->num_frames_decoded++;
  return next_weight_cutoff;
}

/* faster-decoder.cc:41-56 InitDecoding + :120-152 Decode/AdvanceDecoding */
static void decoder_decode(Decoder *d, Decodable *dec) {
  Elem *l = hl_clear(d);
  for (Elem *e = l, *t; e != NULL; e = t) { t = e->tail; hl_delete(d, e); } /* ClearToks */
  int32_t start = d->g->start;
  Token *st = tok_new(d, 0, 0, 0.0f, start, 0.0f, 0, NULL); /* dummy_arc(0,0,One,start) */
  if (!st) return;
  hl_insert(d, start, st);
  process_nonemitting(d, (double)FLT_MAX); /* :53 numeric_limits<float>::max() */
  d->num_frames_decoded = 0;
  while (d->num_frames_decoded < dec->T && !d->oom && !dec->err) {
    double weight_cutoff = process_emitting(d, dec);
    process_nonemitting(d, weight_cutoff);
  }
}

/* faster-decoder.cc:346-353 */
static int reached_final(const Decoder *d) {
  for (const Elem *e = d->list_head; e != NULL; e = e->tail)
    if (e->val->cost != INFINITY && d->g->final[e->key] != INFINITY) return 1;
  return 0;
}

static void decoder_init(Decoder *d, const orc_graph *g, const orc_align_config *cfg,
                         orc_align_stats *stats) {
  memset(d, 0, sizeof(*d));
  d->g = g; d->cfg = *cfg; d->num_frames_decoded = -1; d->stats = stats;
  d->bucket_list_tail = NOBUCKET;
  d->elem_pool.item = sizeof(Elem); d->tok_pool.item = sizeof(Token);
  hl_set_size(d, 1000); /* faster-decoder.cc:30 toks_.SetSize(1000) */
}
static void decoder_free(Decoder *d) {
  pool_free_all(&d->elem_pool); pool_free_all(&d->tok_pool);
  free(d->buckets); free((void *)d->queue); free(d->tmp_array);
}


/* ------------------------------------------------------------------------- */
/* diag-gmm.cc:761-778 MergedComponentsLogdet */
static float merged_logdet(int32_t D, float w1, float w2, const float *f1, const float *f2, const float *s1,
                           const float *s2) {
  float w_sum = w1 + w2, r = w2 / w1, q = w1 / w_sum, acc = 0.0f;
  for (int32_t d = 0; d < D; ++d) {
    float tm = (f1[d] + f2[d] * r) * q;
    float tv = (s1[d] + s2[d] * r) * q - tm * tm;
    acc += logf(tv);
  }
  return (float)(-0.5 * acc); /* :774 double literal x float sum, stored to float */
}

/* diag-gmm.cc:557-759 DiagGmm::Merge */
int orc_diag_gmm_merge(int32_t *G_io, int32_t D, int32_t target, float *weights, float *gconsts, float *miv, float *iv,
                       int32_t *history, int32_t *num_history) {
  const int32_t G = *G_io;
  if (num_history) *num_history = 0;
  if (target <= 0 || G < target) return ORC_ERR_ARG;   /* :558-561 KHG_ERR */
  if (G == target) return ORC_OK;                      /* :563-567 */
  float *vars = (float *)malloc(sizeof(float) * (size_t)G * D), *means = (float *)malloc(sizeof(float) * (size_t)G * D);
  if (!vars || !means) { free(vars); free(means); return ORC_ERR_NOMEM; }
  for (size_t i = 0; i < (size_t)G * D; ++i) {         /* :573-578 / :617-623 */
    vars[i] = 1.0f / iv[i];
    means[i] = miv[i] * vars[i];
    vars[i] = vars[i] + means[i] * means[i];
  }
  if (target == 1) {                                   /* :571-611 global mean and variance */
    float wsum = 0.0f;
    for (int32_t d = 0; d < D; ++d) {
      float a = 0.0f, b = 0.0f;
      for (int32_t g = 0; g < G; ++g) { a += weights[g] * means[(size_t)g * D + d]; b += weights[g] * vars[(size_t)g * D + d]; }
      miv[d] = a; iv[d] = b;
    }
    for (int32_t g = 0; g < G; ++g) wsum += weights[g];
    weights[0] = wsum;
    if (!(fabsf(wsum - 1.0f) <= 1e-6f * (fabsf(wsum) + 1.0f))) {   /* ApproxEqual(w, 1, 1e-6), kaldi-math.h */
      for (int32_t d = 0; d < D; ++d) { miv[d] *= weights[0]; iv[d] *= weights[0]; }   /* :601-603 (as written: times, not divided by) */
      weights[0] = 1.0f;
    }
    for (int32_t d = 0; d < D; ++d) { iv[d] = 1.0f / (iv[d] - miv[d] * miv[d]); miv[d] = miv[d] * iv[d]; }
    *G_io = 1;
    free(vars); free(means);
    return orc_compute_gconsts(1, D, weights, iv, miv, gconsts, NULL);
  }
  char *disc = (char *)calloc((size_t)G, 1);
  float *logdet = (float *)malloc(sizeof(float) * (size_t)G), *dl = (float *)calloc((size_t)G * G, sizeof(float));
  if (!disc || !logdet || !dl) { free(vars); free(means); free(disc); free(logdet); free(dl); return ORC_ERR_NOMEM; }
  for (int32_t g = 0; g < G; ++g) {                    /* :620 logdet = 0.5 * sum log inv_vars */
    float a = 0.0f;
    for (int32_t d = 0; d < D; ++d) a += logf(iv[(size_t)g * D + d]);
    logdet[g] = 0.5f * a;
  }
  for (int32_t i = 0; i < G; ++i)                      /* :636-648 */
    for (int32_t j = 0; j < i; ++j) {
      float w1 = weights[i], w2 = weights[j], w_sum = w1 + w2;
      float ml = merged_logdet(D, w1, w2, means + (size_t)i * D, means + (size_t)j * D, vars + (size_t)i * D, vars + (size_t)j * D);
      dl[(size_t)i * G + j] = w_sum * ml - w1 * logdet[i] - w2 * logdet[j];
    }
  int32_t nh = 0;
  for (int32_t removed = 0; removed < G - target; ++removed) {   /* :651-727 */
    float best = -FLT_MAX;
    int32_t mi = -1, mj = -1;
    for (int32_t i = 0; i < G; ++i) {
      if (disc[i]) continue;
      for (int32_t j = 0; j < i; ++j) {
        if (disc[j]) continue;
        if (dl[(size_t)i * G + j] > best) { best = dl[(size_t)i * G + j]; mi = i; mj = j; }
      }
    }
    if (mi == mj || mi < 0 || mj < 0) { free(vars); free(means); free(disc); free(logdet); free(dl); return ORC_ERR_ARG; }
    if (history) { history[nh++] = mi; history[nh++] = mj; }
    float w1 = weights[mi], w2 = weights[mj], w_sum = w1 + w2, r = w2 / w1;
    float *mI = means + (size_t)mi * D, *mJ = means + (size_t)mj * D, *vI = vars + (size_t)mi * D, *vJ = vars + (size_t)mj * D;
    float ld = 0.0f;
    for (int32_t d = 0; d < D; ++d) {
      mI[d] = (mI[d] + r * mJ[d]) * w1 / w_sum;        /* :680-681 ((a + r b) w1) / w_sum */
      vI[d] = (vI[d] + r * vJ[d]) * w1 / w_sum;
      iv[(size_t)mi * D + d] = 1.0f / (vI[d] - mI[d] * mI[d]);
      miv[(size_t)mi * D + d] = mI[d] * iv[(size_t)mi * D + d];
      ld += logf(iv[(size_t)mi * D + d]);
    }
    weights[mi] = w_sum;
    logdet[mi] = 0.5f * ld;
    disc[mj] = 1;
    for (int32_t j = 0; j < G; ++j) {                  /* :709-726 */
      if (j == mi || disc[j]) continue;
      float a1 = weights[mi], a2 = weights[j], as = a1 + a2;
      float ml = merged_logdet(D, a1, a2, mI, means + (size_t)j * D, vI, vars + (size_t)j * D);
      float t = as * ml - a1 * logdet[mi] - a2 * logdet[j];
      dl[(size_t)mi * G + j] = t; dl[(size_t)j * G + mi] = t;
    }
  }
  int32_t m = 0;                                       /* :729-757 compaction */
  for (int32_t i = 0; i < G; ++i) {
    if (disc[i]) continue;
    weights[m] = weights[i];
    if (m != i) { memmove(miv + (size_t)m * D, miv + (size_t)i * D, sizeof(float) * D); memmove(iv + (size_t)m * D, iv + (size_t)i * D, sizeof(float) * D); }
    ++m;
  }
  *G_io = m;
  if (num_history) *num_history = nh;
  free(vars); free(means); free(disc); free(logdet); free(dl);
  return orc_compute_gconsts(m, D, weights, iv, miv, gconsts, NULL);
}

/* ---- test hooks onto the HashList restatement above (the decoder's own code, no second copy), so that
 * tests/test_oracle_pins.py can replay the reference's csrc/hash-list-test.cc against it.  Values are carried in
 * the Token* slot as integers. ---- */
void *orc_hl_create(void) {
  Decoder *d = (Decoder *)calloc(1, sizeof(Decoder));
  if (!d) return NULL;
  d->bucket_list_tail = NOBUCKET;
  d->elem_pool.item = sizeof(Elem); d->tok_pool.item = sizeof(Token);
  return d;
}
void orc_hl_destroy(void *h) { if (h) { decoder_free((Decoder *)h); free(h); } }
void orc_hl_set_size(void *h, int64_t size) { hl_set_size((Decoder *)h, (size_t)size); }
/* hash-list-inl.h:65-83 Find */
static Elem *hl_find(Decoder *d, int32_t key) {
  size_t index = (size_t)key % d->hash_size;
  Bucket *bucket = &d->buckets[index];
  if (bucket->last_elem == NULL) return NULL;
  Elem *head = (bucket->prev_bucket == NOBUCKET ? d->list_head : d->buckets[bucket->prev_bucket].last_elem->tail),
       *tail = bucket->last_elem->tail;
  for (Elem *e = head; e != tail; e = e->tail)
    if (e->key == key) return e;
  return NULL;
}
int orc_hl_find(void *h, int32_t key, int64_t *val) {
  Elem *e = hl_find((Decoder *)h, key);
  if (!e) return 0;
  if (val) *val = (int64_t)(intptr_t)e->val;
  return 1;
}
/* Find-then-set-or-Insert, the idiom of hash-list-test.cc:31-37 */
void orc_hl_put(void *h, int32_t key, int64_t val) {
  Decoder *d = (Decoder *)h;
  Elem *e = hl_find(d, key);
  if (e) e->val = (Token *)(intptr_t)val;
  else hl_insert(d, key, (Token *)(intptr_t)val);
}
/* Insert proper: returns 1 when a new element was made, 0 when the key existed (its value is left alone, :129-143) */
int orc_hl_insert(void *h, int32_t key, int64_t val) {
  Elem *e = hl_insert((Decoder *)h, key, (Token *)(intptr_t)val);
  return e && (int64_t)(intptr_t)e->val == val && 1;
}
/* GetList (:56-57): the elements in list order */
int64_t orc_hl_list(void *h, int32_t *keys, int64_t *vals, int64_t cap) {
  int64_t n = 0;
  for (const Elem *e = ((Decoder *)h)->list_head; e != NULL; e = e->tail, ++n)
    if (n < cap) { if (keys) keys[n] = e->key; if (vals) vals[n] = (int64_t)(intptr_t)e->val; }
  return n;
}
/* hash-list-test.cc:49-59: h = Clear(); SetSize(new_size); for each old element: Insert(key + shift, val); Delete(old) */
int64_t orc_hl_clear_reinsert(void *h, int64_t new_size, int32_t shift) {
  Decoder *d = (Decoder *)h;
  Elem *e = hl_clear(d), *tmp;
  hl_set_size(d, (size_t)new_size);
  int64_t n = 0;
  for (; e != NULL; e = tmp, ++n) {
    hl_insert(d, e->key + shift, e->val);
    tmp = e->tail;
    hl_delete(d, e);
  }
  return n;
}

/* Clear() and Delete() of every element: what the decoder does with a frame's list once it is expanded (faster-decoder.cc:158, :240) */
void orc_hl_drop(void *h) {
  Decoder *d = (Decoder *)h;
  for (Elem *e = hl_clear(d), *t; e != NULL; e = t) { t = e->tail; hl_delete(d, e); }
}

/* decoder-wrappers.cc:16-108 with faster-decoder.cc:355-423 GetBestPath and kaldifst's
 * GetLinearSymbolSequence (restated from its Kaldi semantics) */
static int align_core(const orc_align_config *cfg, float acoustic_scale, const orc_graph *g,
                      Decodable *dec, int32_t *alignment, int32_t *words, int32_t max_words,
                      int32_t *num_words, float *like, int32_t *status, orc_align_stats *stats) {
  *num_words = 0; *like = 0.0f; *status = ORC_ALIGN_ERROR;
  if ((cfg->retry_beam != 0 && cfg->retry_beam <= cfg->beam) || cfg->beam <= 0.0) return ORC_ERR_ARG;
  if (g->start < 0 || g->num_states == 0) return ORC_OK; /* :35-41 num_error++ */

  Decoder d;
  orc_align_config c = *cfg; /* FasterDecoderOptions decode_opts; decode_opts.beam = config.beam */
  decoder_init(&d, g, &c, stats);
  int rc = ORC_OK, retried = 0;
  decoder_decode(&d, dec);
  int ans = (!d.oom && !dec->err) ? reached_final(&d) : 0;
  if (!d.oom && !dec->err && !ans && cfg->retry_beam != 0.0) { /* :55-67 */
    retried = 1;
    d.cfg.beam = cfg->retry_beam;
    decoder_decode(&d, dec);
    ans = (!d.oom && !dec->err) ? reached_final(&d) : 0;
  }
  if (d.oom) rc = ORC_ERR_NOMEM;
  else if (dec->err) rc = dec->err;
  else if (!ans) {
    *status = ORC_ALIGN_ERROR | (retried ? ORC_ALIGN_RETRIED : 0);
  } else {
    /* GetBestPath(use_final_probs = true), is_final == true branch (:373-382) */
    Token *best_tok = NULL;
    double best_cost = INFINITY;
    for (const Elem *e = d.list_head; e != NULL; e = e->tail) {
      double this_cost = e->val->cost + g->final[e->key];
      if (this_cost < best_cost && this_cost != INFINITY) { best_cost = this_cost; best_tok = e->val; }
    }
    if (!best_tok) {
      *status = ORC_ALIGN_ERROR | (retried ? ORC_ALIGN_RETRIED : 0); /* :82-89 */
    } else {
      /* :390-403 arcs in reverse; the last one is the fake start token and is dropped */
      size_t n = 0;
      for (Token *t = best_tok; t != NULL; t = t->prev) ++n;
      Token **chain = (Token **)malloc(n * sizeof(Token *));
      if (!chain) rc = ORC_ERR_NOMEM;
      else {
        size_t i = n;
        for (Token *t = best_tok; t != NULL; t = t->prev) chain[--i] = t;
        /* LatticeWeight product in path order: Times = (a1+a2, b1+b2) in float */
        float v1 = 0.0f, v2 = 0.0f;
        int32_t na = 0, nw = 0;
        for (i = 1; i < n; ++i) {
          Token *t = chain[i];
          float tot_cost = t->cost - (t->prev ? t->prev->cost : 0.0); /* :393 float(double diff) */
          float graph_cost = t->weight;
          float ac_cost = tot_cost - graph_cost;
          v1 = graph_cost + v1;
          v2 = ac_cost + v2;
          if (t->ilabel != 0) alignment[na++] = t->ilabel;
          if (t->olabel != 0) { if (nw < max_words) words[nw] = t->olabel; ++nw; }
        }
        v1 = g->final[best_tok->nextstate] + v1; /* :415-417 LatticeWeight(final, 0) */
        v2 = 0.0f + v2;
        *num_words = nw;
        *like = -(v1 + v2) / acoustic_scale; /* decoder-wrappers.cc:95 */
        *status = ORC_ALIGN_DONE | (retried ? ORC_ALIGN_RETRIED : 0);
        free(chain);
        (void)na;
      }
    }
  }
  decoder_free(&d);
  return rc;
}

int orc_align_utterance(const orc_align_config *cfg, float acoustic_scale, const orc_graph *g,
                        const orc_model *m, const int32_t *id2pdf, int32_t num_tids, int32_t T,
                        const float *feats, int32_t *alignment, int32_t *words,
                        int32_t max_words, int32_t *num_words, float *like, int32_t *status,
                        orc_align_stats *stats) {
  Decodable dec;
  memset(&dec, 0, sizeof(dec));
  dec.m = m; dec.feats = feats; dec.id2pdf = id2pdf; dec.num_tids = num_tids; dec.T = T;
  dec.scale = acoustic_scale; dec.stats = stats;
  int32_t maxG = 1;
  for (int32_t p = 0; p < m->num_pdfs; ++p) {
    int32_t G = m->gauss_off[p + 1] - m->gauss_off[p];
    if (G > maxG) maxG = G;
  }
  dec.cache_ll = (float *)malloc(sizeof(float) * (size_t)m->num_pdfs);
  dec.cache_time = (int32_t *)malloc(sizeof(int32_t) * (size_t)m->num_pdfs);
  dec.tmp = (float *)malloc(sizeof(float) * (size_t)maxG);
  if (!dec.cache_ll || !dec.cache_time || !dec.tmp) { free(dec.cache_ll); free(dec.cache_time); free(dec.tmp); return ORC_ERR_NOMEM; }
  for (int32_t p = 0; p < m->num_pdfs; ++p) dec.cache_time[p] = -1; /* ResetLogLikeCache */
  int rc = align_core(cfg, acoustic_scale, g, &dec, alignment, words, max_words, num_words, like,
                      status, stats);
  free(dec.cache_ll); free(dec.cache_time); free(dec.tmp);
  return rc;
}

static int build_pdf2col(int32_t npdf, const int32_t *pdfs, int32_t **map, int32_t *n) {
  int32_t maxp = -1;
  for (int32_t j = 0; j < npdf; ++j) if (pdfs[j] > maxp) maxp = pdfs[j];
  *n = maxp + 1;
  *map = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxp + 2));
  if (!*map) return ORC_ERR_NOMEM;
  for (int32_t i = 0; i <= maxp; ++i) (*map)[i] = -1;
  for (int32_t j = 0; j < npdf; ++j) (*map)[pdfs[j]] = j;
  return ORC_OK;
}

int orc_align_utterance_ll(const orc_align_config *cfg, float acoustic_scale, const orc_graph *g,
                           const int32_t *id2pdf, int32_t num_tids, int32_t T, int32_t npdf,
                           const int32_t *pdfs, const float *ll, int64_t ll_stride,
                           int32_t *alignment, int32_t *words, int32_t max_words,
                           int32_t *num_words, float *like, int32_t *status,
                           orc_align_stats *stats) {
  Decodable dec;
  memset(&dec, 0, sizeof(dec));
  dec.ll = ll; dec.ll_stride = ll_stride; dec.id2pdf = id2pdf; dec.num_tids = num_tids; dec.T = T;
  dec.scale = acoustic_scale; dec.stats = stats;
  int rc = build_pdf2col(npdf, pdfs, &dec.pdf2col, &dec.pdf2col_n);
  if (rc) return rc;
  rc = align_core(cfg, acoustic_scale, g, &dec, alignment, words, max_words, num_words, like,
                  status, stats);
  free(dec.pdf2col);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Exact Viterbi (second opinion; not a reference function)                   */
int orc_exact_viterbi_ll(float acoustic_scale, const orc_graph *g, const int32_t *id2pdf,
                         int32_t num_tids, int32_t T, int32_t npdf, const int32_t *pdfs,
                         const float *ll, int64_t ll_stride, int32_t *alignment,
                         double *best_cost_out, int32_t *status) {
  *status = ORC_ALIGN_ERROR; *best_cost_out = INFINITY;
  int32_t S = g->num_states;
  if (g->start < 0 || S == 0) return ORC_OK;
  int32_t A = g->arc_off[S];
  int32_t *pdf2col = NULL, pn = 0;
  int rc = build_pdf2col(npdf, pdfs, &pdf2col, &pn);
  if (rc) return rc;
  double *cur = (double *)malloc(sizeof(double) * (size_t)S);
  double *nxt = (double *)malloc(sizeof(double) * (size_t)S);
  int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)S * (size_t)(T + 1));
  int32_t *src = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A > 0 ? A : 1));
  if (!cur || !nxt || !bp || !src) { free(cur); free(nxt); free(bp); free(src); free(pdf2col); return ORC_ERR_NOMEM; }
  for (int32_t s = 0; s < S; ++s)
    for (int32_t a = g->arc_off[s]; a < g->arc_off[s + 1]; ++a) src[a] = s;
  for (int32_t s = 0; s < S; ++s) { cur[s] = INFINITY; bp[s] = -1; }
  cur[g->start] = 0.0;
  for (int32_t layer = 0; layer <= T && rc == ORC_OK; ++layer) {
    int32_t *bpl = bp + (size_t)layer * S;
    /* epsilon closure on `cur` (Bellman-Ford to the fixpoint) */
    int changed = 1, iters = 0;
    while (changed && iters++ <= S + 1) {
      changed = 0;
      for (int32_t a = 0; a < A; ++a) {
        if (g->ilabel[a] != 0) continue;
        double c = cur[src[a]];
        if (c == INFINITY) continue;
        double nw = c + g->weight[a];
        if (nw < cur[g->nextstate[a]]) { cur[g->nextstate[a]] = nw; bpl[g->nextstate[a]] = a; changed = 1; }
      }
    }
    if (layer == T) break;
    int32_t *bpn = bp + (size_t)(layer + 1) * S;
    for (int32_t s = 0; s < S; ++s) { nxt[s] = INFINITY; bpn[s] = -1; }
    for (int32_t a = 0; a < A; ++a) {
      int32_t tid = g->ilabel[a];
      if (tid == 0) continue;
      double c = cur[src[a]];
      if (c == INFINITY) continue;
      if (tid < 1 || tid > num_tids) { rc = ORC_ERR_ARG; break; }
      int32_t pdf = id2pdf[tid];
      if (pdf < 0 || pdf >= pn || pdf2col[pdf] < 0) { rc = ORC_ERR_ARG; break; }
      float v = ll[(int64_t)pdf2col[pdf] * ll_stride + layer];
      float ac_cost = -1 * (acoustic_scale * v);
      double nw = (c + g->weight[a]) + ac_cost; /* faster-decoder.h:125 */
      if (nw < nxt[g->nextstate[a]]) { nxt[g->nextstate[a]] = nw; bpn[g->nextstate[a]] = a; }
    }
    double *t = cur; cur = nxt; nxt = t;
  }
  if (rc == ORC_OK) {
    int32_t best = -1; double bc = INFINITY;
    for (int32_t s = 0; s < S; ++s) {
      double c = cur[s] + g->final[s];
      if (c < bc && c != INFINITY) { bc = c; best = s; }
    }
    if (best >= 0) {
      int32_t layer = T, s = best; long guard = (long)(T + 1) * (S + 2);
      while (guard-- > 0) {
        int32_t a = bp[(size_t)layer * S + s];
        if (a < 0) break;
        if (g->ilabel[a] != 0) { alignment[layer - 1] = g->ilabel[a]; --layer; }
        s = src[a];
      }
      *best_cost_out = bc;
      *status = ORC_ALIGN_DONE;
    }
  }
  free(cur); free(nxt); free(bp); free(src); free(pdf2col);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* scripts/gmm_acc_stats_ali.py:46-56 driving mle-am-diag-gmm.cc:41-52,
 * mle-diag-gmm.cc:145-158,123-143 and transition-model.h:183-189             */
int orc_acc_stats_ali(const orc_model *m, const int32_t *id2pdf, int32_t num_tids, int32_t T,
                      const float *feats, const int32_t *ali, float weight, orc_accs *accs,
                      double *log_like_out) {
  int32_t D = m->dim, maxG = 1;
  for (int32_t p = 0; p < m->num_pdfs; ++p) {
    int32_t G = m->gauss_off[p + 1] - m->gauss_off[p];
    if (G > maxG) maxG = G;
  }
  float *post = (float *)malloc(sizeof(float) * (size_t)maxG);
  if (!post) return ORC_ERR_NOMEM;
  double log_like = 0.0; /* python float */
  int rc = ORC_OK;
  for (int32_t t = 0; t < T; ++t) {
    int32_t tid = ali[t];
    if (tid < 1 || tid > num_tids) { rc = ORC_ERR_ARG; break; }
    int32_t pdf = id2pdf[tid];
    if (pdf < 0 || pdf >= m->num_pdfs) { rc = ORC_ERR_ARG; break; }
    accs->trans_acc[tid] += 1.0; /* transition-model.h:183-189, prob = 1.0 */
    const float *x = feats + (size_t)t * D;
    float ll;
    rc = orc_component_posteriors(m, pdf, x, post, &ll); /* mle-diag-gmm.cc:152 */
    if (rc) break;
    int32_t g0 = m->gauss_off[pdf], G = m->gauss_off[pdf + 1] - g0;
    for (int32_t g = 0; g < G; ++g) post[g] *= weight; /* :153 */
    for (int32_t g = 0; g < G; ++g) {
      accs->occ[g0 + g] += (double)post[g]; /* :132 */
      double *ma = accs->mean_acc + (size_t)(g0 + g) * D, *va = accs->var_acc + (size_t)(g0 + g) * D;
      for (int32_t d = 0; d < D; ++d) {
        float pm = post[g] * x[d];  /* :135 float product, then widened */
        float xx = x[d] * x[d];
        float pv = post[g] * xx;    /* :138-140 */
        ma[d] += (double)pm;
        va[d] += (double)pv;
      }
    }
    accs->total_log_like += ll * weight; /* mle-am-diag-gmm.cc:49 float product */
    accs->total_frames += weight;        /* :50 */
    log_like += ll;                      /* gmm_acc_stats_ali.py:54 */
  }
  free(post);
  if (log_like_out) *log_like_out = log_like;
  return rc;
}

/* ------------------------------------------------------------------------- */
void orc_mle_opts_default(orc_mle_opts *o) {
  o->min_gaussian_weight = 1.0e-05f; o->min_gaussian_occupancy = 10.0f;
  o->min_variance = 0.001; o->remove_low_count_gaussians = 1; o->variance_floor_vector = NULL;
}

/* model-common.cc:72-85 */
uint16_t orc_augment_gmm_flags(uint16_t flags) {
  if (flags & 0x2) flags |= 0x1;
  if (flags & 0x1) flags |= 0x4;
  if (!(flags & 0x4)) flags |= 0x4;
  return flags;
}

/* mle-diag-gmm.cc:479-499 */
float orc_ml_objective(int32_t G, int32_t D, const float *gconsts, const float *means_invvars,
                       const float *inv_vars, const double *occ, const double *mean_acc,
                       const double *var_acc, uint16_t acc_flags) {
  double dot = 0.0;
  for (int32_t g = 0; g < G; ++g) dot += occ[g] * (double)gconsts[g];
  float obj = dot; /* float obj = double */
  if (acc_flags & 0x1) {
    double s = 0.0;
    for (size_t i = 0; i < (size_t)G * D; ++i) s += mean_acc[i] * (double)means_invvars[i];
    obj += s; /* float += double */
  }
  if (acc_flags & 0x2) {
    double s = 0.0;
    for (size_t i = 0; i < (size_t)G * D; ++i) s += var_acc[i] * (double)inv_vars[i];
    obj -= 0.5 * s;
  }
  return obj;
}

/* mle-diag-gmm.cc:243-390 */
int orc_mle_diag_gmm_update(const orc_mle_opts *o, int32_t *G_io, int32_t D, const double *occ,
                            const double *mean_acc, const double *var_acc, uint16_t acc_flags,
                            uint16_t flags, float *weights, float *gconsts, float *means_invvars,
                            float *inv_vars, float *obj_change, float *count,
                            int32_t *floored_elems, int32_t *floored_gauss, int32_t *removed) {
  int32_t G = *G_io;
  if (flags & ~acc_flags) return ORC_ERR_ARG; /* :252-254 */
  double occ_sum = 0.0;
  for (int32_t g = 0; g < G; ++g) occ_sum += occ[g];
  int32_t elements_floored = 0, gauss_floored = 0, nb;
  int rc = orc_compute_gconsts(G, D, weights, inv_vars, means_invvars, gconsts, &nb); /* :265 */
  if (rc) return rc;
  float obj_old = orc_ml_objective(G, D, gconsts, means_invvars, inv_vars, occ, mean_acc, var_acc, acc_flags);

  /* diag-gmm-normal.cc:14-20 */
  size_t n = (size_t)G * D;
  double *nw = (double *)malloc(sizeof(double) * (size_t)G);
  double *nvars = (double *)malloc(sizeof(double) * n);
  double *nmeans = (double *)malloc(sizeof(double) * n);
  double *old_means = (double *)malloc(sizeof(double) * n); /* oldg in CopyToDiagGmm */
  int32_t *to_remove = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
  double *var = (double *)malloc(sizeof(double) * (size_t)D);
  double *old_mean = (double *)malloc(sizeof(double) * (size_t)D);
  if (!nw || !nvars || !nmeans || !old_means || !to_remove || !var || !old_mean) {
    free(nw); free(nvars); free(nmeans); free(old_means); free(to_remove); free(var); free(old_mean);
    return ORC_ERR_NOMEM;
  }
  for (int32_t g = 0; g < G; ++g) nw[g] = (double)weights[g];
  for (size_t i = 0; i < n; ++i) {
    nvars[i] = 1.0 / (double)inv_vars[i];
    nmeans[i] = (double)means_invvars[i] * nvars[i];
    old_means[i] = nmeans[i];
  }
  int32_t nrem = 0;
  for (int32_t i = 0; i < G; ++i) {
    double oc = occ[i];
    double prob = (occ_sum > 0.0) ? oc / occ_sum : 1.0 / G;
    if (oc > o->min_gaussian_occupancy && prob > o->min_gaussian_weight) { /* :285-286 */
      nw[i] = prob;
      for (int32_t d = 0; d < D; ++d) old_mean[d] = nmeans[(size_t)i * D + d];
      if (acc_flags & (0x1 | 0x2))
        for (int32_t d = 0; d < D; ++d) nmeans[(size_t)i * D + d] = mean_acc[(size_t)i * D + d] / oc;
      if (acc_flags & 0x2) {
        for (int32_t d = 0; d < D; ++d) {
          double v = var_acc[(size_t)i * D + d] / oc;
          double mu = nmeans[(size_t)i * D + d];
          var[d] = v - mu * mu; /* :300-302 */
        }
        if (!(flags & 0x1)) { /* :306-310 */
          for (int32_t d = 0; d < D; ++d) {
            double dm = old_mean[d] - nmeans[(size_t)i * D + d];
            var[d] = var[d] + dm * dm;
          }
        }
        int32_t floored = 0;
        for (int32_t d = 0; d < D; ++d)
          if (o->variance_floor_vector) {              /* :311-322 */
            if (var[d] < o->variance_floor_vector[d]) { var[d] = o->variance_floor_vector[d]; ++floored; }
          } else if (var[d] < o->min_variance) { var[d] = o->min_variance; ++floored; } /* :324-330 */
        if (floored != 0) { elements_floored += floored; ++gauss_floored; }
        for (int32_t d = 0; d < D; ++d) nvars[(size_t)i * D + d] = var[d];
      }
    } else {
      if (o->remove_low_count_gaussians && nrem < G - 1) {
        to_remove[nrem++] = i; /* :343-351 */
      } else {
        double mw = (double)o->min_gaussian_weight;
        nw[i] = prob > mw ? prob : mw; /* :358-359 */
      }
    }
  }
  /* diag-gmm-normal.cc:22-48 CopyToDiagGmm(gmm, flags) */
  if (flags & 0x4) for (int32_t g = 0; g < G; ++g) weights[g] = (float)nw[g];
  if (flags & 0x2) {
    for (size_t i = 0; i < n; ++i) inv_vars[i] = (float)(1.0 / nvars[i]);
    if (!(flags & 0x1))
      for (size_t i = 0; i < n; ++i) means_invvars[i] = (float)old_means[i] * inv_vars[i];
  }
  if (flags & 0x1)
    for (size_t i = 0; i < n; ++i) means_invvars[i] = (float)nmeans[i] * inv_vars[i];

  rc = orc_compute_gconsts(G, D, weights, inv_vars, means_invvars, gconsts, &nb); /* :367 */
  if (rc == ORC_OK) {
    float obj_new = orc_ml_objective(G, D, gconsts, means_invvars, inv_vars, occ, mean_acc, var_acc, acc_flags);
    if (obj_change) *obj_change = obj_new - obj_old;
    if (count) *count = occ_sum; /* float = double */
    if (floored_elems) *floored_elems = elements_floored;
    if (floored_gauss) *floored_gauss = gauss_floored;
    if (nrem > 0) {
      /* diag-gmm.cc:853-938: remove one at a time (indices sorted ascending), renormalising
       * the weights after every removal (new_weights /= new_weights.sum(), float) */
      for (int32_t r = 0; r < nrem; ++r) {
        int32_t gi = to_remove[r] - r;
        for (int32_t g = gi; g < G - 1; ++g) {
          weights[g] = weights[g + 1];
          memcpy(means_invvars + (size_t)g * D, means_invvars + (size_t)(g + 1) * D, sizeof(float) * (size_t)D);
          memcpy(inv_vars + (size_t)g * D, inv_vars + (size_t)(g + 1) * D, sizeof(float) * (size_t)D);
        }
        --G;
        float s = 0.0f;
        for (int32_t g = 0; g < G; ++g) s += weights[g];
        for (int32_t g = 0; g < G; ++g) weights[g] /= s;
      }
      rc = orc_compute_gconsts(G, D, weights, inv_vars, means_invvars, gconsts, &nb); /* :381 */
    }
    if (removed) *removed = nrem;
  }
  *G_io = G;
  free(nw); free(nvars); free(nmeans); free(old_means); free(to_remove); free(var); free(old_mean);
  return rc;
}

/* transition-model.cc:657-750 (share_for_pdfs == false) + :339-359 */
int orc_transition_mle_update(int32_t num_tstates, const int32_t *state2id,
                              const int32_t *self_loop_of, const double *stats, float floor_,
                              float mincount, float *log_probs, float *nsl_log_probs,
                              float *objf_impr, float *count) {
  float count_sum = 0.0f, objf_impr_sum = 0.0f;
  for (int32_t ts = 1; ts <= num_tstates; ++ts) {
    int32_t n = state2id[ts + 1] - state2id[ts];
    if (n < 1) return ORC_ERR_ARG;
    if (n > 1) {
      double tstate_tot = 0;
      for (int32_t k = 0; k < n; ++k) tstate_tot += stats[state2id[ts] + k];
      count_sum += tstate_tot; /* float += double */
      if (tstate_tot < mincount) continue;
      float *new_probs = (float *)malloc(sizeof(float) * (size_t)n);
      float *old_probs = (float *)malloc(sizeof(float) * (size_t)n);
      if (!new_probs || !old_probs) { free(new_probs); free(old_probs); return ORC_ERR_NOMEM; }
      for (int32_t k = 0; k < n; ++k) old_probs[k] = expf(log_probs[state2id[ts] + k]);
      for (int32_t k = 0; k < n; ++k) new_probs[k] = stats[state2id[ts] + k] / tstate_tot;
      for (int32_t it = 0; it < 3; ++it) {
        float s = 0.0f;
        for (int32_t k = 0; k < n; ++k) s += new_probs[k];
        for (int32_t k = 0; k < n; ++k) new_probs[k] /= s;
        for (int32_t k = 0; k < n; ++k) new_probs[k] = new_probs[k] > floor_ ? new_probs[k] : floor_;
      }
      for (int32_t k = 0; k < n; ++k) {
        double objf_change = stats[state2id[ts] + k] * (logf(new_probs[k]) - logf(old_probs[k]));
        objf_impr_sum += objf_change;
      }
      for (int32_t k = 0; k < n; ++k) {
        float lp = logf(new_probs[k]);
        if (lp - lp != 0.0f) { free(new_probs); free(old_probs); return ORC_ERR_NAN; }
        log_probs[state2id[ts] + k] = lp;
      }
      free(new_probs); free(old_probs);
    }
  }
  if (objf_impr) *objf_impr = objf_impr_sum;
  if (count) *count = count_sum;
  /* ComputeDerivedOfProbs :339-359 */
  for (int32_t ts = 1; ts <= num_tstates; ++ts) {
    int32_t tid = self_loop_of[ts];
    if (tid == 0) nsl_log_probs[ts] = 0.0f;
    else {
      float self_loop_prob = expf(log_probs[tid]), non_self_loop_prob = 1.0 - self_loop_prob;
      if (non_self_loop_prob <= 0.0) non_self_loop_prob = 1.0e-10;
      nsl_log_probs[ts] = logf(non_self_loop_prob);
    }
  }
  return ORC_OK;
}

/* ---- utterance-parallel driver for bench.py's cpu_baseline (variant B) ----
 * Not a reference function: the reference is single-threaded.  N POSIX threads take utterances from a shared counter and
 * run exactly what the one-thread baseline runs per utterance (orc_align_utterance, then orc_acc_stats_ali on success),
 * each into its own accumulator set, until the utterances run out or `budget_seconds` have passed.  Graphs arrive as
 * the concatenated CSR of khg_utts_create (weights already include the transition costs). */
#include <pthread.h>
#include <time.h>

typedef struct {
  const orc_align_config *cfg; float acoustic_scale;
  const orc_model *m; const int32_t *id2pdf; int32_t num_tids;
  int32_t first_utt, n_utt;
  const int64_t *frame_off, *state_off, *arc_off;
  const int32_t *start, *ilabel, *olabel, *nextstate;
  const float *weight, *final, *feats;
  double budget_seconds;
  struct timespec t0;
  volatile int32_t next;       /* shared utterance cursor */
  pthread_mutex_t mu;
  pthread_cond_t cv;           /* start gate: accumulators are allocated and touched on every thread before the clock starts */
  int32_t ready_count, go;
  int64_t frames_done; int32_t utts_done; int32_t failed;
  double t_end;                /* when the last thread finished its last utterance */
  /* optional: what the pass computed, kept for parity checks at scale (bench.py `check`) -- written outside the timed work */
  int32_t *ali_out;            /* [frame_off[first_utt + n_utt] - frame_off[first_utt]] transition-ids, 0 on failed utterances */
  int32_t *status_out;         /* [n_utt] ORC_ALIGN_* bits, -1 = not reached within the budget */
  float *like_out;             /* [n_utt] */
  orc_accs *acc_sum;           /* the threads' private accumulators summed (in thread-finish order) */
} orc_mt_job;

static double mt_elapsed(const struct timespec *t0) {
  struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)(t.tv_sec - t0->tv_sec) + 1e-9 * (double)(t.tv_nsec - t0->tv_nsec);
}

static void *mt_worker(void *arg) {
  orc_mt_job *j = (orc_mt_job *)arg;
  const orc_model *m = j->m;
  const int64_t sumG = m->gauss_off[m->num_pdfs];
  orc_accs acc;
  acc.occ = (double *)calloc((size_t)sumG, sizeof(double));
  acc.mean_acc = (double *)calloc((size_t)sumG * m->dim, sizeof(double));
  acc.var_acc = (double *)calloc((size_t)sumG * m->dim, sizeof(double));
  acc.trans_acc = (double *)calloc((size_t)j->num_tids + 1, sizeof(double));
  acc.total_frames = acc.total_log_like = 0.0;
  int64_t frames = 0; int32_t utts = 0, failed = 0;
  const int have = acc.occ && acc.mean_acc && acc.var_acc && acc.trans_acc;
  if (have) {   /* touch every page now: a long-running job pays its page faults once, a few-second sample would be all faults */
    memset(acc.occ, 0, sizeof(double) * (size_t)sumG);
    memset(acc.mean_acc, 0, sizeof(double) * (size_t)sumG * m->dim);
    memset(acc.var_acc, 0, sizeof(double) * (size_t)sumG * m->dim);
  }
  pthread_mutex_lock(&j->mu);
  ++j->ready_count;
  pthread_cond_broadcast(&j->cv);
  while (!j->go) pthread_cond_wait(&j->cv, &j->mu);
  pthread_mutex_unlock(&j->mu);
  if (have) {
    for (;;) {
      if (mt_elapsed(&j->t0) >= j->budget_seconds) break;
      pthread_mutex_lock(&j->mu);
      const int32_t u = j->next < j->n_utt ? j->first_utt + j->next++ : -1;
      pthread_mutex_unlock(&j->mu);
      if (u < 0) break;
      const int64_t s0 = j->state_off[u], S = j->state_off[u + 1] - s0, a0 = j->arc_off[s0];
      const int32_t T = (int32_t)(j->frame_off[u + 1] - j->frame_off[u]);
      int32_t *loc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S + 1));
      int32_t *ali = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T > 0 ? T : 1));
      int32_t *words = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + S + 8));
      if (!loc || !ali || !words) { free(loc); free(ali); free(words); break; }
      for (int64_t s = 0; s <= S; ++s) loc[s] = (int32_t)(j->arc_off[s0 + s] - a0);
      orc_graph g = {(int32_t)S, j->start[u], loc, j->ilabel + a0, j->olabel + a0, j->weight + a0, j->nextstate + a0, j->final + s0};
      int32_t nw = 0, status = 0; float like = 0.0f;
      const float *x = j->feats + j->frame_off[u] * m->dim;
      int rc = orc_align_utterance(j->cfg, j->acoustic_scale, &g, m, j->id2pdf, j->num_tids, T, x, ali, words, (int32_t)(T + S + 8), &nw,
                                   &like, &status, NULL);
      if (rc == ORC_OK && (status & ORC_ALIGN_ERROR) == 0) {
        double ll = 0.0;
        orc_acc_stats_ali(m, j->id2pdf, j->num_tids, T, x, ali, 1.0f, &acc, &ll);
      } else {
        ++failed;
      }
      if (j->status_out) j->status_out[u - j->first_utt] = rc == ORC_OK ? status : (ORC_ALIGN_ERROR | 0x4000);
      if (j->like_out) j->like_out[u - j->first_utt] = like;
      if (j->ali_out && T > 0) {
        int32_t *dst = j->ali_out + (j->frame_off[u] - j->frame_off[j->first_utt]);
        if (rc == ORC_OK && (status & ORC_ALIGN_ERROR) == 0) memcpy(dst, ali, sizeof(int32_t) * (size_t)T);
        else memset(dst, 0, sizeof(int32_t) * (size_t)T);
      }
      frames += T; ++utts;
      free(loc); free(ali); free(words);
    }
  }
  const double t_end = mt_elapsed(&j->t0);
  pthread_mutex_lock(&j->mu);
  if (have && j->acc_sum) {    /* after the clock: the cross-thread sum (AccumAmDiagGmm::Add, mle-am-diag-gmm.cc:119-128) */
    orc_accs *s = j->acc_sum;
    for (int64_t i = 0; i < sumG; ++i) s->occ[i] += acc.occ[i];
    for (int64_t i = 0; i < sumG * m->dim; ++i) { s->mean_acc[i] += acc.mean_acc[i]; s->var_acc[i] += acc.var_acc[i]; }
    for (int32_t i = 0; i <= j->num_tids; ++i) s->trans_acc[i] += acc.trans_acc[i];
    s->total_frames += acc.total_frames; s->total_log_like += acc.total_log_like;
  }
  pthread_mutex_unlock(&j->mu);
  free(acc.occ); free(acc.mean_acc); free(acc.var_acc); free(acc.trans_acc);
  pthread_mutex_lock(&j->mu);
  j->frames_done += frames; j->utts_done += utts; j->failed += failed;
  if (t_end > j->t_end) j->t_end = t_end;
  pthread_mutex_unlock(&j->mu);
  return NULL;
}

int orc_em_pass_mt(const orc_align_config *cfg, float acoustic_scale, const orc_model *m, const int32_t *id2pdf, int32_t num_tids,
                   int32_t first_utt, int32_t n_utt, const int64_t *frame_off, const float *feats, const int64_t *state_off,
                   const int32_t *start, const int64_t *arc_off, const int32_t *ilabel, const int32_t *olabel,
                   const float *weight, const int32_t *nextstate, const float *final, int32_t num_threads,
                   double budget_seconds, int64_t *frames_done, int32_t *utts_done, int32_t *failed, double *seconds) {
  return orc_em_pass_mt_keep(cfg, acoustic_scale, m, id2pdf, num_tids, first_utt, n_utt, frame_off, feats, state_off, start, arc_off, ilabel,
                             olabel, weight, nextstate, final, num_threads, budget_seconds, frames_done, utts_done, failed, seconds, NULL,
                             NULL, NULL, NULL);
}

int orc_em_pass_mt_keep(const orc_align_config *cfg, float acoustic_scale, const orc_model *m, const int32_t *id2pdf, int32_t num_tids,
                        int32_t first_utt, int32_t n_utt, const int64_t *frame_off, const float *feats, const int64_t *state_off,
                        const int32_t *start, const int64_t *arc_off, const int32_t *ilabel, const int32_t *olabel,
                        const float *weight, const int32_t *nextstate, const float *final, int32_t num_threads,
                        double budget_seconds, int64_t *frames_done, int32_t *utts_done, int32_t *failed, double *seconds,
                        int32_t *ali_out, int32_t *status_out, float *like_out, orc_accs *acc_sum) {
  if (num_threads < 1 || n_utt < 0) return ORC_ERR_ARG;
  orc_mt_job j;
  memset(&j, 0, sizeof(j));
  j.ali_out = ali_out; j.status_out = status_out; j.like_out = like_out; j.acc_sum = acc_sum;
  if (status_out) for (int32_t i = 0; i < n_utt; ++i) status_out[i] = -1;
  j.cfg = cfg; j.acoustic_scale = acoustic_scale; j.m = m; j.id2pdf = id2pdf; j.num_tids = num_tids;
  j.first_utt = first_utt; j.n_utt = n_utt; j.frame_off = frame_off; j.state_off = state_off; j.arc_off = arc_off;
  j.start = start; j.ilabel = ilabel; j.olabel = olabel; j.nextstate = nextstate; j.weight = weight; j.final = final; j.feats = feats;
  j.budget_seconds = budget_seconds;
  pthread_mutex_init(&j.mu, NULL);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)num_threads);
  if (!th) return ORC_ERR_NOMEM;
  pthread_cond_init(&j.cv, NULL);
  int started = 0;
  for (int i = 0; i < num_threads; ++i) { if (pthread_create(&th[i], NULL, mt_worker, &j) != 0) break; ++started; }
  pthread_mutex_lock(&j.mu);
  while (j.ready_count < started) pthread_cond_wait(&j.cv, &j.mu);
  clock_gettime(CLOCK_MONOTONIC, &j.t0);       /* the clock starts when every thread is ready */
  j.go = 1;
  pthread_cond_broadcast(&j.cv);
  pthread_mutex_unlock(&j.mu);
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  pthread_cond_destroy(&j.cv);
  if (seconds) *seconds = j.t_end;
  free(th);
  pthread_mutex_destroy(&j.mu);
  if (frames_done) *frames_done = j.frames_done;
  if (utts_done) *utts_done = j.utts_done;
  if (failed) *failed = j.failed;
  return started > 0 ? ORC_OK : ORC_ERR_NOMEM;
}
