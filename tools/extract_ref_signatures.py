#!/usr/bin/env python3
"""Extracts the INTERFACE SCHEMA of the reference's pybind11 surface for the hot path and writes it
to tests/golden/pybind_signatures.json.

Runs in the BUILD container only (it reads /root/reference/kaldi-hmm-gmm/python/csrc/*.cc, which does not exist on
the GPU box); the JSON it writes is data -- class / method / property names, keyword-argument names, which arguments
carry defaults -- not source text.  tests/test_pybind_signatures.py checks the package against it on the CPU.

What is parsed, per in-scope binding file:
  py::class_<...>(*m, "Name")  followed by its chain of  .def / .def_static / .def_property[_readonly] /
      .def_readwrite / .def_readonly  calls            -> classes[Name].members
  py::enum_<...>(*m, "Name").value("k", ...)           -> enums[Name]
  m->def("name", ...)                                  -> functions[name]
  scripts/gmm_*.py: def name(args...)                  -> scripts[] (ast: argument names and defaults only)
and inside each call the top-level  py::arg("x")  /  py::arg("x") = default  entries, in order.
"""
import json
import os
import re
import sys

REF = "/root/reference/kaldi-hmm-gmm/python/csrc"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pybind_signatures.json")

# binding files on the hot path and its §8(f) widening; everything else is listed with the reason it is not a row
IN_SCOPE = ["model-common.cc", "diag-gmm.cc", "am-diag-gmm.cc", "mle-diag-gmm.cc", "mle-am-diag-gmm.cc",
            "decodable-itf.cc", "decodable-am-diag-gmm.cc", "decoder-wrappers.cc", "faster-decoder.cc", "hmm-utils.cc",
            "hmm-topology.cc", "transition-information.cc", "transition-model.cc", "context-dep.cc",
            "training-graph-compiler.cc"]
OUT_OF_SCOPE = {
    "add-self-loops.cc": "graph construction inside kaldifst-based H transducer building (SURVEY §2 out of scope)",
    "cluster-utils.cc": "tree building / clustering (SURVEY §2 out of scope)",
    "clusterable-classes.cc": "tree building / clustering (SURVEY §2 out of scope)",
    "decodable-ctc.cc": "CTC decodable, not on the HMM-GMM training path",
    "determinize-lattice-pruned.cc": "lattice determinization (decode-time, SURVEY §2 out of scope)",
    "eigen-test.cc": "binding self-test of Eigen conversions (its stored vectors are used as oracle pins instead)",
    "event-map.cc": "decision-tree event maps (SURVEY §2 out of scope)",
    "lattice-faster-decoder.cc": "lattice-generating decoder (decode-time, SURVEY §2 out of scope)",
    "lattice-simple-decoder.cc": "lattice-generating decoder (decode-time, SURVEY §2 out of scope)",
    "tree-renderer.cc": "graphviz rendering of trees",
}
# in-scope FILES may still bind names outside §8 (they need kaldifst lattice decoders / H transducers / clustering)
OUT_OF_SCOPE_NAMES = {
    "decode_utterance_lattice_simple": "needs LatticeSimpleDecoder (out of scope)",
    "decode_utterance_lattice_faster": "needs LatticeFasterDecoder (out of scope)",
    "HTransducerConfig": "H-transducer construction via kaldifst (out of scope; §8f-1 builds linear graphs directly)",
    "get_h_transducer": "H-transducer construction via kaldifst (out of scope)",
    "DiagGmm.merge_kmeans": "needs ClusterKMeans (cluster-utils, out of scope)",
    "DiagGmm.merge_kmeans:cfg": "needs ClusterKMeansOptions",
}


def strip_comments(s):
    out, i, n = [], 0, len(s)
    while i < n:
        c = s[i]
        if c == '"':
            j = i + 1
            while j < n and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            out.append(s[i:j + 1]); i = j + 1
        elif s.startswith("//", i):
            j = s.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif s.startswith("/*", i):
            j = s.find("*/", i) + 2
            out.append(re.sub(r"[^\n]", " ", s[i:j])); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def match_paren(s, i):
    """s[i] == '(' -> index of the matching ')' (strings and char literals skipped)."""
    d, n = 0, len(s)
    while i < n:
        c = s[i]
        if c == '"':
            i += 1
            while s[i] != '"':
                i += 2 if s[i] == "\\" else 1
        elif c == "'":
            i += 1
            while s[i] != "'":
                i += 2 if s[i] == "\\" else 1
        elif c == "(":
            d += 1
        elif c == ")":
            d -= 1
            if d == 0:
                return i
        i += 1
    raise ValueError("unbalanced")


def drop_braces(s):
    """removes {...} bodies (lambda bodies) so that only the call's own argument list remains."""
    out, d, i = [], 0, 0
    while i < len(s):
        c = s[i]
        if c == '"':
            j = i + 1
            while s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            if d == 0:
                out.append(s[i:j + 1])
            i = j + 1
            continue
        if c == "{":
            d += 1
        elif c == "}":
            d -= 1
        elif d == 0:
            out.append(c)
        i += 1
    return "".join(out)


ARG_RE = re.compile(r'py::arg\(\s*"(\w+)"\s*\)')


def parse_args(call):
    """call = text inside .def( ... ) with lambda bodies removed -> [{name, has_default, default}]"""
    args = []
    for m in ARG_RE.finditer(call):
        j = m.end()
        while j < len(call) and call[j].isspace():
            j += 1
        default = None
        if j < len(call) and call[j] == "=":
            k, d = j + 1, 0
            while k < len(call):
                if call[k] in "(<":
                    d += 1
                elif call[k] in ")>":
                    if d == 0:
                        break
                    d -= 1
                elif call[k] == "," and d == 0:
                    break
                k += 1
            default = " ".join(call[j + 1:k].split())
        args.append({"name": m.group(1), "has_default": default is not None, "default": default})
    return args


def lambda_arity(call):
    """number of parameters of the bound lambda, `self` excluded (None when a member pointer is bound: arity unknown)."""
    m = re.search(r"\[\s*\]\s*\(", call)
    if not m:
        return None
    op = m.end() - 1
    inner = call[op + 1:match_paren(call, op)]
    parts, d, cur = [], 0, ""
    for c in inner:
        if c in "(<[":
            d += 1
        elif c in ")>]":
            d -= 1
        if c == "," and d == 0:
            parts.append(cur); cur = ""
        else:
            cur += c
    if cur.strip():
        parts.append(cur)
    if parts and re.search(r"\bself\b", parts[0]):
        parts = parts[1:]
    return len(parts)


def first_token(call):
    call = call.lstrip()
    if call.startswith('"'):
        return call[1:call.index('"', 1)]
    if call.startswith("py::init"):
        return "__init__"
    if call.startswith("py::pickle"):
        return "__pickle__"
    return call.split("(")[0].split(",")[0].strip()


def lineno(text, pos):
    return text.count("\n", 0, pos) + 1


def parse_chain(text, pos):
    """pos = just after the closing ')' of py::class_<...>(...): consumes `.defXXX( ... )` calls until ';'."""
    members = []
    while True:
        m = re.compile(r"\s*\.\s*(\w+)\s*\(").match(text, pos)
        if not m:
            break
        op = text.index("(", m.end() - 1)
        cl = match_paren(text, op)
        members.append((m.group(1), text[op + 1:cl], lineno(text, m.start(1))))
        pos = cl + 1
    return members, pos


def parse_file(fname, schema):
    with open(os.path.join(REF, fname)) as f:
        text = strip_comments(f.read())
    # `using PyClass = X;` is only documentation here: the python name is the string literal.
    for m in re.finditer(r"py::(class_|enum_)\s*<", text):
        # skip the template argument list (balanced <>), then the constructor call (*m, "Name" ...)
        i, d = m.end() - 1, 0
        while True:
            if text[i] == "<":
                d += 1
            elif text[i] == ">":
                d -= 1
                if d == 0:
                    break
            i += 1
        tmpl = " ".join(text[m.end():i].split())
        op = text.index("(", i)
        cl = match_paren(text, op)
        name = re.search(r'"(\w+)"', text[op:cl]).group(1)
        chain, _ = parse_chain(text, cl + 1)
        if m.group(1) == "enum_":
            vals = [first_token(c) for k, c, _ in chain if k == "value"]
            schema["enums"][name] = {"file": fname, "line": lineno(text, m.start()), "values": vals,
                                     "export_values": any(k == "export_values" for k, _, _ in chain)}
            continue
        bases = [b.strip() for b in tmpl.split(",")[1:]]
        cls = schema["classes"].setdefault(name, {"file": fname, "line": lineno(text, m.start()), "bases": bases, "members": []})
        for kind, call, ln in chain:
            if not kind.startswith("def"):
                continue
            flat = drop_braces(call)
            cls["members"].append({"name": first_token(flat), "kind": kind, "line": ln, "args": parse_args(flat),
                                   "lambda_arity": lambda_arity(call) if kind == "def" else None})
    for m in re.finditer(r"\bm\s*->\s*def\s*\(", text):
        op = m.end() - 1
        cl = match_paren(text, op)
        flat = drop_braces(text[op + 1:cl])
        schema["functions"].append({"name": first_token(flat), "file": fname, "line": lineno(text, m.start()), "args": parse_args(flat),
                                    "lambda_arity": lambda_arity(text[op + 1:cl])})


SCRIPTS = ["gmm_init_mono.py", "gmm_align_compiled.py", "gmm_acc_stats_ali.py", "gmm_est.py", "gmm_boost_silence.py", "gmm_info.py"]


def parse_scripts(schema):
    """the module-level functions of the reference's scripts/gmm_*.py the EM driver calls: argument names + defaults"""
    import ast
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(REF))), "scripts")
    for fname in SCRIPTS:
        with open(os.path.join(root, fname)) as f:
            tree = ast.parse(f.read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and not node.name.startswith("_") and node.name != "main":
                a = node.args
                names = [x.arg for x in a.args]
                nd = len(a.defaults)
                args = [{"name": n, "has_default": i >= len(names) - nd,
                         "default": ast.unparse(a.defaults[i - (len(names) - nd)]) if i >= len(names) - nd else None}
                        for i, n in enumerate(names)]
                schema["scripts"].append({"name": node.name, "file": "scripts/" + fname, "line": node.lineno, "args": args})


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference is not present (this script runs in the build container only)")
    present = sorted(f for f in os.listdir(REF) if f.endswith(".cc") and f != "kaldi-hmm-gmm.cc")
    unknown = [f for f in present if f not in IN_SCOPE and f not in OUT_OF_SCOPE]
    if unknown:
        sys.exit("binding files neither in scope nor waived: %s" % unknown)
    schema = {"generated_by": "tools/extract_ref_signatures.py", "reference_dir": "kaldi-hmm-gmm/python/csrc",
              "in_scope_files": IN_SCOPE, "out_of_scope_files": OUT_OF_SCOPE, "out_of_scope_names": OUT_OF_SCOPE_NAMES,
              "classes": {}, "enums": {}, "functions": [], "scripts": []}
    for f in IN_SCOPE:
        parse_file(f, schema)
    parse_scripts(schema)
    with open(OUT, "w") as f:
        json.dump(schema, f, indent=1, sort_keys=True)
        f.write("\n")
    nm = sum(len(c["members"]) for c in schema["classes"].values())
    print("%d classes (%d members), %d enums, %d functions, %d script functions -> %s" % (
        len(schema["classes"]), nm, len(schema["enums"]), len(schema["functions"]), len(schema["scripts"]), os.path.normpath(OUT)))


if __name__ == "__main__":
    main()
