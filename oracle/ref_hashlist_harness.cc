// TEST INFRASTRUCTURE (oracle/): a driver around the REFERENCE's own HashList -- the header is compiled where it lies,
// /root/reference/kaldi-hmm-gmm/csrc/hash-list.h (+ hash-list-inl.h, stl-utils.h, log.h: standard library only), nothing is
// copied.  Built by oracle/Makefile into oracle/_ref/hashlist_ref when /root/reference is present.  It pins the list-order
// semantics the order-faithful decoder depends on (buckets in order of first occupation, insertion order inside a bucket,
// SetSize only between lists): tests/golden/make_hashlist_golden.py records its answers, tests/test_oracle_pins.py replays
// them against oracle/khg_oracle.c's restatement (and against this binary itself when it exists).
//
// Protocol (stdin, one command per line; answers on stdout):
//   S n          SetSize(n)
//   I key val    Insert(key, val)                         -> "I 1" (new element) | "I 0" (key was there, value untouched)
//   P key val    Find(key) ? set its value : Insert       (the idiom of csrc/hash-list-test.cc:31-37)
//   F key        Find(key)                                -> "F val" | "F none"
//   L            GetList()                                -> "L key:val key:val ..."
//   D            e = Clear(); Delete every element        (what the decoder does with a frame's list once it is expanded)
//   R n shift    e = Clear(); SetSize(n); re-Insert every element as (key + shift, val), Delete the old one
//                (csrc/hash-list-test.cc:49-59)           -> "R count"
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>

#include "kaldi-hmm-gmm/csrc/hash-list.h"

int main() {
  khg::HashList<int32_t, int64_t> h;
  std::string op;
  while (std::cin >> op) {
    if (op == "S") { long long n; std::cin >> n; h.SetSize((size_t)n); }
    else if (op == "I") {
      long long k, v; std::cin >> k >> v;
      auto* e = h.Insert((int32_t)k, (int64_t)v);
      std::printf("I %d\n", e->val == v ? 1 : 0);
    } else if (op == "P") {
      long long k, v; std::cin >> k >> v;
      auto* e = h.Find((int32_t)k);
      if (e) e->val = v; else h.Insert((int32_t)k, (int64_t)v);
    } else if (op == "F") {
      long long k; std::cin >> k;
      auto* e = h.Find((int32_t)k);
      if (e) std::printf("F %lld\n", (long long)e->val); else std::printf("F none\n");
    } else if (op == "L") {
      std::printf("L");
      for (auto* e = h.GetList(); e != nullptr; e = e->tail) std::printf(" %d:%lld", (int)e->key, (long long)e->val);
      std::printf("\n");
    } else if (op == "D") {
      for (auto* e = h.Clear(); e != nullptr;) { auto* t = e->tail; h.Delete(e); e = t; }
    } else if (op == "R") {
      long long n, shift; std::cin >> n >> shift;
      auto* e = h.Clear();
      h.SetSize((size_t)n);
      long long cnt = 0;
      for (decltype(e) tmp; e != nullptr; e = tmp, ++cnt) {
        h.Insert(e->key + (int32_t)shift, e->val);
        tmp = e->tail;
        h.Delete(e);
      }
      std::printf("R %lld\n", cnt);
    } else { std::fprintf(stderr, "unknown command %s\n", op.c_str()); return 2; }
  }
  // leave the list empty so that ~HashList's leak check stays quiet
  for (auto* e = h.Clear(); e != nullptr;) { auto* t = e->tail; h.Delete(e); e = t; }
  return 0;
}
