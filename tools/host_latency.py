"""Host-side time of each library call of one EM pass at the bench shape (the GPU is idle when the pass starts: what the
first step of a timed region pays before its first kernel can start)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from kaldi_hmm_gmm_amd import Context, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
import ctypes as C
U = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, G, D = synth.CONFIGS["tri5000x64"]
m = synth.make_model(P, G, D, seed=1)
gc = np.zeros(m.weights.shape[0], np.float32)
_lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float), _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
ut = synth.make_utts(m, U, seed=3, feats=False)
dev = torch.device("cuda", 0)
feats = synth.sample_feats_torch(m, ut.frame_pdf, 5, dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = Context(0, stream=st.cuda_stream)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf); tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
us = UtteranceSet(ctx, tm, ut.frame_off, (feats.data_ptr(), feats), dim=D, graphs=ut.graphs)
accs = DeviceAccs(ctx, dm, tm)
for it in range(10):
    torch.cuda.synchronize(); ctx.sync()
    idle = 0.0
    time.sleep(idle)
    if it >= 5: ctx.set_timing(True)
    t = [time.perf_counter()]
    accs.zero(); t.append(time.perf_counter())
    us.loglikes(dm, reachable_only=True); t.append(time.perf_counter())
    us.align(tm, beam=200.0, retry_beam=0.0, acoustic_scale=0.1, download=False); t.append(time.perf_counter())
    us.acc_stats(dm, tm, accs); t.append(time.perf_counter())
    torch.cuda.synchronize(); ctx.sync(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    if it >= 5: ctx.timings(); ctx.set_timing(False)
    print(f"pass {it} (timing {'on' if it >= 5 else 'off'}): zero {d[0]:.2f}  loglikes {d[1]:.2f}  align {d[2]:.2f}  acc_stats {d[3]:.2f}  drain {d[4]:.2f}  total {(t[-1] - t[0]) * 1e3:.2f} ms")
