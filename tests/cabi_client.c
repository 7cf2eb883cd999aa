/* tests/cabi_client.c -- a plain C11 caller of include/khg_hip.h (no Python, no torch, no C++): what a
 * maintainer's FFI layer sees.  One EM pass over a tiny synthetic problem through the C-ABI
 *   khg_compute_gconsts -> khg_model_create / khg_tm_create / khg_utts_create -> khg_loglikes_reachable ->
 *   khg_align -> khg_acc_stats -> khg_model_mle_update -> khg_model_download
 * checked against the CPU oracle (oracle/khg_oracle.h; test infrastructure): alignments bit-exact, statistics
 * rtol 2e-5, M-step parameters bit-exact / gconsts <= 4 ulp on the device's own statistics.
 * `cabi_client --no-gpu` runs only the host-side entry points and checks that khg_ctx_create fails loudly.
 * Built by __graft_entry__.build() with gcc -std=c11 -Wall -Wextra -Werror -pedantic. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/khg_hip.h"
#include "../oracle/khg_oracle.h"

#define NPHONE 4
#define P (3 * NPHONE)
#define G 3
#define D 8
#define U 6
#define MAXL 5

static uint64_t rng_state = 20230414u;
static double urand(void) {   /* SplitMix64 -> [0, 1) */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2.0 * log(1.0 - urand())) * cos(6.283185307179586 * urand()); }

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, khg_last_error()); return 1; } \
  } while (0)
#define REQUIRE(cond)                                                                \
  do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static double ulps(float a, float b) {
  float m = fabsf(a) > fabsf(b) ? fabsf(a) : fabsf(b);
  float sp = nextafterf(m, INFINITY) - m;
  return fabs((double)a - (double)b) / (double)sp;
}

int main(int argc, char **argv) {
  const int no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
  const int sumG = P * G, num_tids = 2 * P;
  /* ---- model ---- */
  int32_t gauss_off[P + 1];
  static float w[P * G], iv[P * G * D], miv[P * G * D], gc[P * G], gc_o[P * G], mean[P * G * D];
  for (int p = 0; p <= P; ++p) gauss_off[p] = p * G;
  for (int p = 0; p < P; ++p) {
    float s = 0;
    for (int g = 0; g < G; ++g) { w[p * G + g] = (float)(0.5 + urand()); s += w[p * G + g]; }
    for (int g = 0; g < G; ++g) w[p * G + g] /= s;
  }
  for (int i = 0; i < sumG * D; ++i) {
    mean[i] = (float)(3.0 * nrand());
    const float var = (float)(0.5 + 1.5 * urand());
    iv[i] = 1.0f / var;
    miv[i] = mean[i] * iv[i];
  }
  int32_t nbad = 0;
  CHECK(khg_compute_gconsts(P, D, gauss_off, w, iv, miv, gc, &nbad));
  for (int p = 0; p < P; ++p) REQUIRE(orc_compute_gconsts(G, D, w + p * G, iv + p * G * D, miv + p * G * D, gc_o + p * G, &nbad) == ORC_OK);
  REQUIRE(memcmp(gc, gc_o, sizeof(gc)) == 0);   /* host ComputeGconsts == oracle, bit for bit */
  /* ---- transition tables of a 3-state left-to-right model: tid 2p+1 self-loop, 2p+2 forward ---- */
  int32_t id2pdf[2 * P + 1], id2state[2 * P + 1];
  uint8_t is_self_loop[2 * P + 1];
  float log_probs[2 * P + 1], nsl[P + 1], cost[2 * P + 1];
  id2pdf[0] = 0; id2state[0] = 0; is_self_loop[0] = 0; log_probs[0] = 0; nsl[0] = 0;
  for (int t = 1; t <= num_tids; ++t) {
    id2pdf[t] = (t - 1) / 2; id2state[t] = id2pdf[t] + 1; is_self_loop[t] = (uint8_t)(t & 1);
    log_probs[t] = logf((t & 1) ? 0.75f : 0.25f);
  }
  for (int s = 1; s <= P; ++s) nsl[s] = logf(1.0f - expf(logf(0.75f)));
  CHECK(khg_scaled_trans_cost(num_tids, log_probs, nsl, id2state, is_self_loop, 1.0f, 0.1f, cost));
  if (no_gpu) {
    khg_ctx *c = NULL;
    const int rc = khg_ctx_create(0, NULL, &c);
    if (rc == 0) { printf("CABI_CLIENT_SKIP: a GPU is present\n"); khg_ctx_destroy(c); return 0; }
    REQUIRE(strlen(khg_last_error()) > 0);
    printf("CABI_CLIENT_OK (host entry points; khg_ctx_create refused: %s)\n", khg_last_error());
    return 0;
  }
  /* ---- utterances: linear graphs of 3L emitting states + a final one ---- */
  int64_t frame_off[U + 1] = {0}, state_off[U + 1] = {0};
  static int64_t arc_off[U * (3 * MAXL + 1) + 1];
  static int32_t start[U], il[U * 6 * MAXL], ol[U * 6 * MAXL], ns[U * 6 * MAXL];
  static float aw[U * 6 * MAXL], fin[U * (3 * MAXL + 1)];
  static float feats[U * 3 * MAXL * 12 * D];
  int64_t nst = 0, narc = 0, nfr = 0;
  arc_off[0] = 0;
  for (int u = 0; u < U; ++u) {
    const int L = 3 + (int)(urand() * (MAXL - 2));
    int cur = 0;
    start[u] = 0;
    for (int k = 0; k < 3 * L; ++k) {
      if (k % 3 == 0) cur = (int)(urand() * NPHONE);
      const int pdf = 3 * cur + k % 3;
      il[narc] = 2 * pdf + 1; ol[narc] = 0; aw[narc] = 0.0f; ns[narc] = k; ++narc;           /* self-loop */
      il[narc] = 2 * pdf + 2; ol[narc] = (k % 3 == 2) ? cur + 1 : 0; aw[narc] = 0.0f; ns[narc] = k + 1; ++narc;
      fin[nst] = INFINITY;
      arc_off[++nst] = narc;
      const int dur = 2 + (int)(urand() * 6);
      const int g = (int)(urand() * G);
      for (int f = 0; f < dur; ++f) {
        for (int d = 0; d < D; ++d) {
          const int row = (pdf * G + g) * D + d;
          feats[nfr * D + d] = mean[row] + (float)(nrand() / sqrt((double)iv[row]));
        }
        ++nfr;
      }
    }
    fin[nst] = 0.0f;            /* the final state, no arcs */
    arc_off[++nst] = narc;
    state_off[u + 1] = nst;
    frame_off[u + 1] = nfr;
  }
  /* ---- the HIP path ---- */
  khg_ctx *ctx; khg_model *m; khg_tm *tm; khg_utts *us; khg_accs *acc;
  CHECK(khg_ctx_create(0, NULL, &ctx));
  CHECK(khg_model_create(ctx, P, D, gauss_off, gc, miv, iv, &m));
  CHECK(khg_tm_create(ctx, num_tids, id2pdf, &tm));
  CHECK(khg_tm_set_trans_cost(tm, cost));
  CHECK(khg_utts_create(ctx, tm, U, D, frame_off, feats, NULL, state_off, start, arc_off, il, ol, aw, ns, fin, &us));
  CHECK(khg_loglikes_reachable(ctx, m, us));
  khg_align_config cfg; khg_align_config_default(&cfg);
  cfg.beam = 200.0f; cfg.retry_beam = 0.0f; cfg.acoustic_scale = 0.1f;
  static int32_t ali[U * 3 * MAXL * 12], words[U * 3 * MAXL * 12 + 64], status[U];
  static int64_t words_off[U + 1];
  static float like[U];
  CHECK(khg_align(ctx, tm, us, &cfg, ali, words, words_off, (int64_t)(sizeof(words) / sizeof(words[0])), like, status));
  CHECK(khg_accs_create(ctx, m, tm, &acc));
  CHECK(khg_acc_stats(ctx, m, tm, us, 1.0f, acc));
  int64_t nacc = 0;
  CHECK(khg_accs_size(acc, &nacc));
  REQUIRE(nacc == sumG * (1 + 2 * D) + num_tids + 1 + 8);
  double *blk = (double *)malloc(sizeof(double) * (size_t)nacc);
  CHECK(khg_accs_download(ctx, acc, blk));
  /* ---- the oracle on the same inputs ---- */
  orc_model om = {P, D, gauss_off, gc, miv, iv};
  static double o_occ[P * G], o_mean[P * G * D], o_var[P * G * D], o_tr[2 * P + 1];
  orc_accs oa = {o_occ, o_mean, o_var, o_tr, 0.0, 0.0};
  orc_align_config oc; orc_align_config_default(&oc);
  oc.beam = 200.0f; oc.retry_beam = 0.0f;
  for (int u = 0; u < U; ++u) {
    const int64_t s0 = state_off[u], S = state_off[u + 1] - s0, a0 = arc_off[s0], A = arc_off[s0 + S] - a0;
    const int T = (int)(frame_off[u + 1] - frame_off[u]);
    int32_t loc_off[3 * MAXL + 2];
    float wt[6 * MAXL];
    for (int s = 0; s <= S; ++s) loc_off[s] = (int32_t)(arc_off[s0 + s] - a0);
    memcpy(wt, aw + a0, sizeof(float) * (size_t)A);
    REQUIRE(orc_add_transition_probs((int32_t)A, il + a0, wt, num_tids, log_probs, nsl, id2state, is_self_loop, 1.0f, 0.1f, 0, NULL) == ORC_OK);
    orc_graph og = {(int32_t)S, 0, loc_off, il + a0, ol + a0, wt, ns + a0, fin + s0};
    int32_t o_ali[3 * MAXL * 12], o_words[64], o_nw = 0, o_status = 0;
    float o_like = 0;
    REQUIRE(orc_align_utterance(&oc, 0.1f, &og, &om, id2pdf, num_tids, T, feats + frame_off[u] * D, o_ali, o_words, 64, &o_nw,
                                &o_like, &o_status, NULL) == ORC_OK);
    REQUIRE(o_status == ORC_ALIGN_DONE && (status[u] & 1) == 0);
    REQUIRE(memcmp(o_ali, ali + frame_off[u], sizeof(int32_t) * (size_t)T) == 0);           /* alignment: bit-exact */
    REQUIRE(words_off[u + 1] - words_off[u] == o_nw && memcmp(o_words, words + words_off[u], sizeof(int32_t) * (size_t)o_nw) == 0);
    REQUIRE(fabsf(like[u] - o_like) <= 2e-5f * fabsf(o_like) + 1e-4f);
    double ll = 0;
    REQUIRE(orc_acc_stats_ali(&om, id2pdf, num_tids, T, feats + frame_off[u] * D, o_ali, 1.0f, &oa, &ll) == ORC_OK);
  }
  const double *d_occ = blk, *d_mean = blk + sumG, *d_var = blk + sumG + sumG * D, *d_tr = blk + sumG + 2 * sumG * D;
  for (int i = 0; i < sumG; ++i) REQUIRE(fabs(d_occ[i] - o_occ[i]) <= 2e-5 * fabs(o_occ[i]) + 1e-6);
  double mx = 0;
  for (int i = 0; i < sumG * D; ++i) if (fabs(o_var[i]) > mx) mx = fabs(o_var[i]);
  for (int i = 0; i < sumG * D; ++i) {
    REQUIRE(fabs(d_mean[i] - o_mean[i]) <= 2e-5 * fabs(o_mean[i]) + 2e-6 * mx);
    REQUIRE(fabs(d_var[i] - o_var[i]) <= 2e-5 * fabs(o_var[i]) + 2e-6 * mx);
  }
  for (int t = 0; t <= num_tids; ++t) REQUIRE(d_tr[t] == o_tr[t]);                          /* counts: exact */
  /* ---- M-step on the device vs the oracle's MleDiagGmmUpdate on the device's own statistics ---- */
  khg_mle_options mo; khg_mle_options_default(&mo);
  mo.min_gaussian_occupancy = 3.0f;
  float objf = 0, cnt = 0; int32_t fe = 0, fg = 0, rm = 0;
  CHECK(khg_model_set_weights(ctx, m, w));
  /* C1 from plain C: the library's own RCCL communicator, here over the one rank this process is; the
   * all-reduce runs on the context's stream between K3 and K4 and must leave the one-rank block as it was */
  {
    char id[KHG_COMM_ID_BYTES];
    void *comm = NULL;
    CHECK(khg_comm_unique_id(id));
    CHECK(khg_comm_create(ctx, 1, 0, id, &comm));
    REQUIRE(comm != NULL);
    CHECK(khg_accs_allreduce(ctx, acc, comm));
    double *blk2 = (double *)malloc(sizeof(double) * (size_t)nacc);
    CHECK(khg_accs_download(ctx, acc, blk2));
    REQUIRE(memcmp(blk, blk2, sizeof(double) * (size_t)nacc) == 0);
    free(blk2);
    CHECK(khg_comm_destroy(comm));
    CHECK(khg_accs_allreduce(ctx, acc, NULL));            /* one-rank job without a communicator: a no-op */
  }
  CHECK(khg_model_mle_update(ctx, m, acc, &mo, 7, &objf, &cnt, &fe, &fg, &rm));
  int64_t newG = 0; int32_t new_off[P + 1];
  CHECK(khg_model_num_gauss(m, &newG, new_off));
  static float nw[P * G], ngc[P * G], nmiv[P * G * D], niv[P * G * D];
  CHECK(khg_model_download(ctx, m, nw, ngc, nmiv, niv));
  orc_mle_opts oo; orc_mle_opts_default(&oo);
  oo.min_gaussian_occupancy = 3.0f;
  int tot_removed = 0; double worst = 0;
  for (int p = 0; p < P; ++p) {
    int32_t Gp = G, ofe, ofg, orm; float oobj, ocnt;
    float pw[G], pgc[G], pmiv[G * D], piv[G * D];
    memcpy(pw, w + p * G, sizeof(pw)); memcpy(pmiv, miv + p * G * D, sizeof(pmiv)); memcpy(piv, iv + p * G * D, sizeof(piv));
    REQUIRE(orc_mle_diag_gmm_update(&oo, &Gp, D, d_occ + p * G, d_mean + p * G * D, d_var + p * G * D, 0xF, 7, pw, pgc, pmiv, piv,
                                    &oobj, &ocnt, &ofe, &ofg, &orm) == ORC_OK);
    tot_removed += orm;
    REQUIRE(new_off[p + 1] - new_off[p] == Gp);
    REQUIRE(memcmp(pw, nw + new_off[p], sizeof(float) * (size_t)Gp) == 0);                 /* bit-exact */
    REQUIRE(memcmp(piv, niv + (size_t)new_off[p] * D, sizeof(float) * (size_t)Gp * D) == 0);
    REQUIRE(memcmp(pmiv, nmiv + (size_t)new_off[p] * D, sizeof(float) * (size_t)Gp * D) == 0);
    for (int g = 0; g < Gp; ++g) { const double e = ulps(pgc[g], ngc[new_off[p] + g]); if (e > worst) worst = e; }
  }
  REQUIRE(worst <= 4.0 && tot_removed == rm && newG == sumG - rm);
  REQUIRE(fabs((double)cnt - (double)nfr) <= 1e-4 * (double)nfr);
  free(blk);
  CHECK(khg_accs_destroy(acc)); CHECK(khg_utts_destroy(us)); CHECK(khg_tm_destroy(tm)); CHECK(khg_model_destroy(m));
  CHECK(khg_ctx_destroy(ctx));
  printf("CABI_CLIENT_OK: %d utterances, %lld frames aligned bit-exactly; M-step removed %d Gaussians, gconsts within %.1f ulp\n",
         U, (long long)nfr, rm, worst);
  return 0;
}
