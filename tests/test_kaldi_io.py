"""Kaldi-format I/O of the objects either side of the hot path: ContextDependency /
EventMap, HmmTopology, TransitionModel -- text and binary round trips, the reference's golden text dump
of a transition model (python/tests/test_transition_model.py:185-230) read back, a hand-written
context-dependent tree file, and pickles (python/csrc/context-dep.cc:64-79)."""
import pickle
import struct

import numpy as np
import pytest

TOPO = """
 <Topology>
 <TopologyEntry>
 <ForPhones> 1 </ForPhones>
 <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
 <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
 <State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
 <State> 3 <PdfClass> 3 <Transition> 3 0.5 <Transition> 4 0.5 </State>
 <State> 4 <PdfClass> 4 <Transition> 4 0.5 <Transition> 5 0.5 </State>
 <State> 5 </State>
 </TopologyEntry>
 <TopologyEntry>
 <ForPhones> 2 3 4 </ForPhones>
 <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
 <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
 <State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
 <State> 3 </State>
 </TopologyEntry>
 </Topology>
"""

# python/tests/test_transition_model.py:185-230 (the reference's own text Write of the model above)
GOLDEN_TM = """<TransitionModel>
<Topology>
<TopologyEntry>
<ForPhones>
1
</ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
<State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
<State> 3 <PdfClass> 3 <Transition> 3 0.5 <Transition> 4 0.5 </State>
<State> 4 <PdfClass> 4 <Transition> 4 0.5 <Transition> 5 0.5 </State>
<State> 5 </State>
</TopologyEntry>
<TopologyEntry>
<ForPhones>
2 3 4
</ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
<State> 2 <PdfClass> 2 <Transition> 2 0.5 <Transition> 3 0.5 </State>
<State> 3 </State>
</TopologyEntry>
</Topology>
<Triples> 14
1 0 0
1 1 1
1 2 2
1 3 3
1 4 4
2 0 5
2 1 6
2 2 7
3 0 8
3 1 9
3 2 10
4 0 11
4 1 12
4 2 13
</Triples>
<LogProbs>
 [ 0 """ + " ".join(["-0.693147"] * 28) + """ ]
</LogProbs>
</TransitionModel>
"""


def _objs():
    import kaldi_hmm_gmm_amd as khg

    topo = khg.HmmTopology()
    topo.read(TOPO)
    tree = khg.monophone_context_dependency(phones=topo.phones, phone2num_pdf_classes=topo.get_phone_to_num_pdf_classes())
    tm = khg.TransitionModel(tree, topo)
    return khg, topo, tree, tm


def test_monophone_tree_structure_and_text():
    khg, topo, tree, tm = _objs()
    # csrc/build-tree-utils.cc:77-93: a table on the phone (key P = 0) of tables on the pdf-class (key -1)
    txt = str(tree)
    assert txt.split()[:6] == ["ContextDependency", "1", "0", "ToPdf", "TE", "0"]
    assert "TE -1 5 ( CE 0 CE 1 CE 2 CE 3 CE 4 )" in " ".join(txt.split())
    assert "NULL" in txt.split()                                   # phone 0 has no entry
    assert txt.split()[-1] == "EndContextDependency"
    assert tree.num_pdfs == 14
    assert tree.compute(phone_seq=[2], pdf_class=1) == (True, 6)   # python/tests/test_context_dep.py:86-114
    assert tree.compute([1], 5) == (False, -1)
    info = tree.get_pdf_info(topo.phones, topo.get_phone_to_num_pdf_classes())
    assert info[0] == [(1, 0)] and info[5] == [(2, 0)] and info[6] == [(2, 1)]


@pytest.mark.parametrize("binary", [False, True])
def test_round_trips(tmp_path, binary):
    khg, topo, tree, tm = _objs()
    f = str(tmp_path / "tree")
    tree.write(binary=binary, filename=f)
    raw = open(f, "rb").read()
    assert (raw[:2] == b"\0B") == binary
    t2 = khg.ContextDependency()
    t2.read(f)
    assert str(t2) == str(tree) and t2.num_pdfs == tree.num_pdfs
    # transition model (with trained, non-trivial probabilities)
    stats = np.zeros(tm.num_transition_ids + 1)
    stats[1:] = np.random.default_rng(0).integers(5, 50, tm.num_transition_ids)
    tm.mle_update(stats, khg.MleTransitionUpdateConfig())
    f2 = str(tmp_path / "tm")
    tm.write(binary=binary, filename=f2)
    tm2 = khg.TransitionModel()
    tm2.read(f2)
    assert [str(t) for t in tm2.tuples] == [str(t) for t in tm.tuples]
    assert tm2.id2pdf_id == tm.id2pdf_id and tm2.state2id == tm.state2id
    tol = 0 if binary else 2e-6            # text carries 6 significant digits
    np.testing.assert_allclose(tm2.log_probs, tm.log_probs, rtol=0, atol=tol * 10 if not binary else 0)
    np.testing.assert_allclose(tm2.non_self_loop_log_probs, tm.non_self_loop_log_probs, rtol=0, atol=1e-5 if not binary else 0)
    # topology alone
    f3 = str(tmp_path / "topo")
    topo.write(binary=binary, filename=f3)
    tp2 = khg.HmmTopology()
    tp2.read_file(f3)
    assert str(tp2) == str(topo)


def test_reads_the_reference_golden_text_dump():
    khg, topo, tree, tm = _objs()
    from kaldi_hmm_gmm_amd import kaldi_io

    tm2 = khg.TransitionModel()
    tm2._read(kaldi_io.Reader(GOLDEN_TM.encode("ascii"), False))
    assert tm2.num_transition_ids == 28 and tm2.num_pdfs == 14
    assert tm2.id2pdf_id == tm.id2pdf_id
    np.testing.assert_allclose(tm2.log_probs, tm.log_probs, atol=1e-6)
    assert " ".join(str(tm2).split()) == " ".join(GOLDEN_TM.split())


def test_binary_layout_is_kaldi_basic_types():
    khg, topo, tree, tm = _objs()
    raw = tree.to_bytes(True)
    assert raw.startswith(b"ContextDependency " + struct.pack("<bi", 4, 1) + struct.pack("<bi", 4, 0) + b"ToPdf TE ")
    i = raw.index(b"TE ") + 3
    assert raw[i: i + 5] == struct.pack("<bi", 4, 0)               # key = P = 0 (signed -> size byte +4)
    assert raw[i + 5: i + 10] == struct.pack("<bI", -4, 5)         # table size, unsigned -> size byte -4


def test_context_dependent_tree_file(tmp_path):
    """A hand-written triphone-style tree (N=3, P=1): split on the left context for phone 1."""
    import kaldi_hmm_gmm_amd as khg

    txt = ("ContextDependency 3 1 ToPdf TE 1 3 ( NULL SE 0 [ 1 ]\n{ TE -1 2 ( CE 0 CE 1 )\n TE -1 2 ( CE 2 CE 3 )\n } \n"
           "TE -1 2 ( CE 4 CE 5 )\n )\n EndContextDependency ")
    f = tmp_path / "tree3"
    f.write_text(txt)
    t = khg.ContextDependency()
    t.read(str(f))
    assert (t.context_width, t.central_position, t.num_pdfs) == (3, 1, 6)
    assert t.compute([1, 1, 2], 1) == (True, 1)        # left context 1 -> "yes" branch
    assert t.compute([2, 1, 2], 1) == (True, 3)
    assert t.compute([2, 2, 1], 0) == (True, 4)
    assert t.compute([1, 3, 1], 0) == (False, -1)
    info = t.get_pdf_info([1, 2], [-1, 2, 2])
    assert info[0] == [(1, 0)] and info[2] == [(1, 0)] and info[5] == [(2, 1)]
    data = pickle.dumps(t, 2)
    assert str(pickle.loads(data)) == str(t)
    # a transition model over the context-dependent tree: phone 1 / state 0 has two tuples (pdf 0 and 2)
    topo = khg.HmmTopology()
    topo.read("<Topology> <TopologyEntry> <ForPhones> 1 2 </ForPhones> <State> 0 <PdfClass> 0 <Transition> 0 0.5 <Transition> 1 0.5 "
              "</State> <State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State> <State> 2 </State> "
              "</TopologyEntry> </Topology>")
    tm = khg.TransitionModel(t, topo)
    assert tm.num_pdfs == 6 and tm.num_transition_states == 6 and tm.num_transition_ids == 12
