#!/usr/bin/env python3
"""Records tests/golden/hashlist_ref.json from the REFERENCE's own HashList: oracle/_ref/hashlist_ref is
/root/reference/kaldi-hmm-gmm/csrc/hash-list.h (+ -inl.h) compiled as it lies behind oracle/ref_hashlist_harness.cc
(`make -C oracle ref`; only possible where /root/reference exists).  Each case is a command script for that harness and the
lines it answered; tests/test_oracle_pins.py replays the scripts against oracle/khg_oracle.c's restatement.

Cases: (a) random Insert / put / Find / list / clear-and-reinsert mixes in the style of csrc/hash-list-test.cc; (b) the decoder's
own usage pattern (faster-decoder.cc:154-240, 337-344): per frame Clear, SetSize(max(size, hash_ratio x tokens)) -- the size never
shrinks --, then Inserts of next-states in arc order with repeats, on state sets LARGER than the hash size, so that buckets are
shared and the list order (buckets in order of first occupation, insertion order inside a bucket) differs from insertion order.
Run from the repo root:  python tests/golden/make_hashlist_golden.py"""
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "oracle", "_ref", "hashlist_ref")


def squeeze(line):
    """A list answer ("L key:val ...") of more than 12 elements is stored as its length and a digest: the order is still pinned."""
    if line.startswith("L") and line.count(":") > 12:
        return "L# %d %s" % (line.count(":"), hashlib.sha1(line.encode()).hexdigest()[:16])
    return line


def run(script):
    r = subprocess.run([BIN], input="\n".join(script) + "\n", capture_output=True, text=True, check=True)
    return [squeeze(x) for x in r.stdout.splitlines()]


def random_mix(rng, n_ops, key_mod, size):
    s = [f"S {size}"]
    for _ in range(n_ops):
        c = rng.random()
        k = int(rng.integers(0, key_mod))
        if c < 0.45:
            s.append(f"I {k} {int(rng.integers(0, 1000))}")
        elif c < 0.6:
            s.append(f"P {k} {int(rng.integers(0, 1000))}")
        elif c < 0.8:
            s.append(f"F {k}")
        elif c < 0.95:
            s.append("L")
        else:
            s.append(f"R {int(rng.integers(1, 3 * key_mod))} {int(rng.integers(0, 5))}")
    s.append("L")
    return s


def decoder_pattern(rng, n_states, n_frames, hash_ratio=2.0, start_size=1000):
    """Tokens spread over a graph of n_states states with random numbering; per frame: the list is cleared (R with shift 0 would
    re-insert: here a fresh frame re-inserts the successors instead), the hash grows to hash_ratio x the token count."""
    perm = rng.permutation(n_states)
    s = [f"S {start_size}"]
    size = start_size
    alive = {0}
    for f in range(n_frames):
        new_sz = int(np.float32(len(alive)) * np.float32(hash_ratio))
        # Clear + SetSize + nothing re-inserted: "R size 0" re-inserts the old elements, so empty the list first by a clear with
        # the harness's R on an EMPTY list -- the list is emptied by re-inserting into a fresh frame below
        if new_sz > size:
            size = new_sz
        s.append(f"X {size}")                     # placeholder, replaced below: Clear (dropping the elements) + SetSize
        nxt = []
        for st in sorted(alive, key=lambda x: int(perm[x])):         # some list order of the frame's tokens
            for d in (0, 1, int(rng.integers(1, 4))):                # self-loop, forward, a skip: repeats are the rule
                if st + d < n_states and rng.random() < 0.9:
                    nxt.append(st + d)
        for st in nxt:
            s.append(f"I {int(perm[st])} {f}")
        s.append("L")
        alive = set(nxt) or {0}
        if len(alive) > 700:
            alive = set(sorted(alive)[-700:])
    return s


def pack_script(script):
    """consecutive "I key val" commands of one value -> "I* val key key ..." (the decoder pattern inserts a frame's states with val = frame)"""
    out, i = [], 0
    while i < len(script):
        a = script[i].split()
        if a[0] == "I":
            j, keys = i, []
            while j < len(script) and script[j].split()[0] == "I" and script[j].split()[2] == a[2]:
                keys.append(script[j].split()[1]); j += 1
            if len(keys) > 3:
                out.append("I* " + a[2] + " " + " ".join(keys)); i = j; continue
        out.append(script[i]); i += 1
    return out


def pack_answers(ans):
    """runs of "I 0" / "I 1" -> "I= 0110..." """
    out, i = [], 0
    while i < len(ans):
        if ans[i] in ("I 0", "I 1"):
            j = i
            while j < len(ans) and ans[j] in ("I 0", "I 1"):
                j += 1
            out.append("I= " + "".join(x[2] for x in ans[i:j])); i = j
        else:
            out.append(ans[i]); i += 1
    return out


def main():
    rng = np.random.default_rng(20230418)
    cases = []
    for i in range(12):
        key_mod = int(rng.choice([7, 50, 200, 1000]))
        size = int(rng.choice([1, 3, 10, 97, 1000]))
        cases.append(random_mix(rng, 250, key_mod, size))
    for n_states, n_frames in ((300, 30), (1500, 50), (2600, 60), (5000, 70)):
        cases.append(decoder_pattern(rng, n_states, n_frames))
    out = []
    for sc in cases:
        # "X size": drop the list and set the size = Clear + Delete all + SetSize; the harness's R with a size re-inserts, so the
        # drop is spelled as: R size 0 on a list we first empty by reading nothing -- simplest is a dedicated pair of commands
        script = []
        for line in sc:
            if line.startswith("X "):
                script.append("D")
                script.append("S " + line.split()[1])
            else:
                script.append(line)
        out.append({"script": pack_script(script), "answers": pack_answers(run(script))})
    with open(os.path.join(ROOT, "tests", "golden", "hashlist_ref.json"), "w") as fh:
        json.dump({"source": "oracle/_ref/hashlist_ref = /root/reference/kaldi-hmm-gmm/csrc/hash-list.h compiled as it lies "
                             "(oracle/ref_hashlist_harness.cc, make -C oracle ref)", "cases": out}, fh, separators=(",", ":"))
    print(len(out), "cases,", sum(len(c["script"]) for c in out), "commands,", sum(len(c["answers"]) for c in out), "answers")


if __name__ == "__main__":
    main()
