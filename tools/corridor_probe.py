"""How much K1 time a CORRIDOR around the alignment would save (experiment, needs tools/bin/libkhg_dbgband.so copied over the library:
khg_dbg_set_band overrides the per-(utterance, pdf) first / last needed frames).  Bands = [first frame - D, last frame + D] of the
frames the generating alignment gives the pdf; K1 alone is timed (the alignment that follows is not valid without K2 masking)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch
from kaldi_hmm_gmm_amd import Context, DeviceModel, DeviceTransitions, UtteranceSet, synth, _lib
from oracle import oracle as orc   # gconsts only
P, G, D = 5000, 64, 40
U = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = synth.make_model(P, G, D, seed=20230418)
gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
ctx = Context(0)
dm = DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
tm = DeviceTransitions(ctx, m.id2pdf)
tm.set_trans_cost(np.zeros(m.num_tids + 1, np.float32))
ut = synth.make_utts(m, U, seed=20230418 + 1000, feats=False)
feats = synth.sample_feats_torch(m, ut.frame_pdf, 7, torch.device("cuda", 0))
us = UtteranceSet(ctx, tm, ut.frame_off, (feats.data_ptr(), feats), dim=D, graphs=ut.graphs)
poff, pdfs = us.pdf_lists()
poff = np.asarray(poff); pdfs = np.asarray(pdfs)
N = int(ut.frame_off[-1])
utt_of_frame = np.repeat(np.arange(U), np.diff(ut.frame_off))
t_in_utt = np.arange(N) - np.repeat(ut.frame_off[:-1], np.diff(ut.frame_off))
key_list = np.repeat(np.arange(U), np.diff(poff)).astype(np.int64) * P + pdfs
assert (np.diff(key_list) > 0).all()
entry = np.searchsorted(key_list, utt_of_frame.astype(np.int64) * P + ut.frame_pdf)
first = np.full(len(pdfs), 2**30, np.int64); last = np.full(len(pdfs), -1, np.int64)
np.minimum.at(first, entry, t_in_utt); np.maximum.at(last, entry, t_in_utt)
T_of_entry = np.repeat(np.diff(ut.frame_off), np.diff(poff))
lib = C.CDLL(_lib.LIB_PATH)
lib.khg_dbg_set_band.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
def k1(label):
    for rep in range(2):
        ctx.set_timing(rep == 1)
        us.loglikes(dm, band=True); ctx.sync()
    km = dict(ctx.timings()); ctx.set_timing(False)
    print(label, "k1 %.2f ms" % km["k1_loglikes"], flush=True)
k1("graph band (product)")
for delta in (64, 32, 16, 8, 0):
    f = np.maximum(first - delta, 0).astype(np.int32); l = np.minimum(last + delta, T_of_entry - 1).astype(np.int32)
    tiles = ((l // 32) - (f // 32) + 1).sum()
    lib.khg_dbg_set_band(C.c_void_p(us.h), f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p))
    k1(f"corridor +-{delta}: {tiles / len(f):.2f} tiles per (utt, pdf)")
