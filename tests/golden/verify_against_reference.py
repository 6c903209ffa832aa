#!/usr/bin/env python3
"""Ties tests/golden/*.json to the reference's own test source, mechanically.

The fixtures were transcribed by hand (build_fixtures.py); a typo there would silently re-pin the oracle.  This
script re-reads the cited `#[test]` functions of the reference checkout as TEXT (nothing of the reference is
compiled, imported or executed), collects the string literals of each function body, and checks that every
input / expected / contains / token string of every fixture entry is found in the test it cites:

  exact    the fixture string equals one literal of the test body (after Rust escape processing; `indoc!`
           literals are dedented the way the macro does it), or
  lines    every non-empty line of the fixture string is a whole line of one of the body's literals (inputs the
           test assembles from several literals, `.trim()`-ed indoc blocks); literals that follow one another with
           only commas and whitespace between them (`concat!`) also count joined together,
  word     (expected tokens only) the token is a whitespace-separated word of one of the literals: the tokenizer
           tests assert counts and slice identity, not token text,
  derived  listed in DERIVED below with the reason and the literals the string is built from (all of which must
           be in the test).

It only runs where /root/reference exists (the build container); tests/test_oracle_golden.py calls it there and
skips elsewhere.  Usage: python tests/golden/verify_against_reference.py [reference_root]
"""
import json
import os
import re
import sys
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- a small reader of Rust source text --------------------------------------------------------------
def _unescape(body):
    """Value of a normal (non-raw) Rust string literal body."""
    out = []
    i, n = 0, len(body)
    while i < n:
        c = body[i]
        if c != "\\":
            out.append(c)
            i += 1
            continue
        i += 1
        e = body[i]
        if e == "n":
            out.append("\n")
        elif e == "t":
            out.append("\t")
        elif e == "r":
            out.append("\r")
        elif e == "0":
            out.append("\0")
        elif e in "\\\"'":
            out.append(e)
        elif e == "x":
            out.append(chr(int(body[i + 1:i + 3], 16)))
            i += 2
        elif e == "u":
            j = body.index("}", i)
            out.append(chr(int(body[i + 2:j].replace("_", ""), 16)))
            i = j
        elif e == "\n":   # line continuation: the newline and the following whitespace are skipped
            i += 1
            while i < n and body[i] in " \t\n\r":
                i += 1
            continue
        else:
            raise ValueError("unknown escape \\%s" % e)
        i += 1
    return "".join(out)


def scan(src, start, stop):
    """(literals, end) of src[start:stop]: every string literal outside comments, with the macro name in front of
    it when it is the first thing inside `name! {` / `name!(`."""
    lits = []
    spans = []   # (begin, end) of every literal, for the adjacency rule
    i = start
    while i < stop:
        c = src[i]
        if src.startswith("//", i):
            i = src.index("\n", i) if "\n" in src[i:] else stop
        elif src.startswith("/*", i):
            i = src.index("*/", i) + 2
        elif c == "'":   # char literal or lifetime
            m = re.match(r"'(\\u\{[0-9a-fA-F_]+\}|\\x[0-9a-fA-F]{2}|\\.|[^\\'])'", src[i:])
            if m:
                lits.append(("char", _unescape(m.group(1))))
                spans.append((i, i + m.end()))
                i += m.end()
            else:
                i += 1
        elif c == "r" and re.match(r'r#*"', src[i:]) and not (i > 0 and (src[i - 1].isalnum() or src[i - 1] == "_")):
            m = re.match(r'r(#*)"', src[i:])
            close = '"' + m.group(1)
            j = src.index(close, i + m.end())
            lits.append((_macro_before(src, i), src[i + m.end():j]))
            spans.append((i, j + len(close)))
            i = j + len(close)
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            lits.append((_macro_before(src, i), _unescape(src[i + 1:j])))
            spans.append((i, j + 1))
            i = j + 1
        else:
            i += 1
    # runs of string literals separated by nothing but commas and whitespace, joined (concat!("a", "b"))
    run = []
    for k, (macro, val) in enumerate(list(lits)):
        if macro == "char":
            run = []
            continue
        if run and re.fullmatch(r"[\s,]*", src[spans[k - 1][1]:spans[k][0]]):
            run.append(val)
        else:
            run = [val]
        if len(run) > 1:
            lits.append(("joined", "".join(run)))
    return lits


def _macro_before(src, i):
    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)!\s*[\(\{\[]\s*$", src[max(0, i - 80):i])
    return m.group(1) if m else ""


def _match_brace(src, i):
    """Index just past the brace that closes the one at src[i], skipping literals and comments."""
    depth, j = 0, i
    while True:
        c = src[j]
        if src.startswith("//", j):
            j = src.index("\n", j)
            continue
        if src.startswith("/*", j):
            j = src.index("*/", j) + 2
            continue
        if c == '"':
            j += 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
        elif c == "r" and re.match(r'r#+"', src[j:]):
            mm = re.match(r'r(#+)"', src[j:])
            j = src.index('"' + mm.group(1), j + mm.end()) + len(mm.group(1))
        elif c == "'":
            mm = re.match(r"'(\\u\{[0-9a-fA-F_]+\}|\\x[0-9a-fA-F]{2}|\\.|[^\\'])'", src[j:])
            if mm:
                j += mm.end() - 1
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1


def test_function(src, line):
    """(name, begin, end) of the body of the function that the 1-based `line` belongs to: the line of its
    `#[test]` attribute, of its `fn`, or any line of its body."""
    lines = src.split("\n")
    starts = [0]
    for l in lines:
        starts.append(starts[-1] + len(l) + 1)
    pos = starts[min(line + 2, len(lines))]   # an attribute line sits at most two lines above its `fn`
    best = None
    for m in re.finditer(r"\bfn\s+([A-Za-z_0-9]+)\s*(?:<[^>]*>)?\s*\(", src[:pos]):
        best = m
    if best is None:
        raise ValueError("no fn at or before line %d" % line)
    i = src.index("{", best.end())
    j = _match_brace(src, i)
    if not (starts[line - 1] < j):
        raise ValueError("line %d is not inside fn %s" % (line, best.group(1)))
    return best.group(1), i, j


def indoc(s):
    """What the indoc! macro makes of its literal: common indentation removed, the newline after the opening
    quote dropped."""
    if s.startswith("\n"):
        s = s[1:]
    lines = s.split("\n")
    ind = [len(l) - len(l.lstrip(" \t")) for l in lines if l.strip()]
    k = min(ind) if ind else 0
    return "\n".join(l[k:] if l.strip() else l.lstrip(" \t") for l in lines)


class Body:
    def __init__(self, root, ref):
        path, line = ref.rsplit(":", 1)
        with open(os.path.join(root, path), encoding="utf-8") as f:
            src = f.read()
        self.name, a, b = test_function(src, int(line))
        self.literals = []
        for macro, val in scan(src, a, b):
            self.literals.append(indoc(val) if macro == "indoc" else val)
        self.lines = set()
        for v in self.literals:
            for l in v.split("\n"):
                self.lines.add(l)
                self.lines.add(l.strip())
        self.numbers = set(re.findall(r"\b\d+(?:\.\d+)?\b", src[a:b]))

    def has(self, s):
        """'exact', 'lines' or None."""
        if s in self.literals or s in (v.strip() for v in self.literals):
            return "exact"
        parts = [l for l in s.split("\n") if l.strip()]
        if parts and all(l in self.lines or l.strip() in self.lines for l in parts):
            return "lines"
        return None

    def has_word(self, s):
        return any(s in v.split() for v in self.literals)


# ---- the fixtures ------------------------------------------------------------------------------------
def strings_of(entry):
    """(label, string) pairs of one fixture entry that must be found in the cited test."""
    out = []
    for key in ("input", "expected", "text", "prefix"):
        v = entry.get(key)
        if isinstance(v, str) and v != "":
            out.append((key, v))
    for key in ("contains", "not_contains", "tokens", "texts"):
        for v in entry.get(key) or []:
            if isinstance(v, str) and v != "":
                out.append((key, v))
    return out


# Strings the cited test does not hold as literals, with the reason (checked by hand against the cited line).
DERIVED = {
    # ref -> {label: (reason, literals of the test the string is built from)}
    "src/sampler/tests.rs:566": {"prefix": ("KvpPrefixSampler renders the variant (\"long\", \"a b c d e f g h i\") as "
                                            "\"meta: long=a b c d e f g h i\" (src/kvp.rs:216, :230)",
                                            ["long", "a b c d e f g h i"])},
    "src/sampler/tests.rs:600": {"prefix": ("KvpPrefixSampler renders the variant (\"source\", \"unit\") as "
                                            "\"meta: source=unit\" (src/kvp.rs:216, :230)", ["source", "unit"])},
    "src/sampler/tests.rs:10637": {"text": ("the test builds the query as (0..n).map(|i| format!(\"word{i}\")) joined "
                                            "by spaces", ["word{i}", " "])},
}


def verify(root):
    problems, checked, modes = [], 0, {"exact": 0, "lines": 0, "word": 0, "derived": 0}
    bodies = {}

    def body(ref):
        if ref not in bodies:
            bodies[ref] = Body(root, ref)
        return bodies[ref]

    def check(fixture, entry):
        nonlocal checked
        ref = entry.get("ref")
        if not ref:
            return
        try:
            b = body(ref)
        except Exception as e:   # noqa: BLE001 - any parse problem is a finding
            problems.append("%s %s: cannot read the cited test: %s" % (fixture, ref, e))
            return
        for label, s in strings_of(entry):
            checked += 1
            mode = b.has(s)
            if mode is None and label == "tokens" and b.has_word(s):
                mode = "word"
            if mode is None and label in DERIVED.get(ref, {}):
                if all(piece in b.literals for piece in DERIVED[ref][label][1]):
                    mode = "derived"
            if mode is None:
                problems.append("%s %s (fn %s): %s %r not found among the test's literals" % (fixture, ref, b.name, label, s[:70]))
            else:
                modes[mode] += 1

    for name in sorted(os.listdir(HERE)):
        if not name.endswith(".json") or name == "example_dataset.json":
            continue
        with open(os.path.join(HERE, name), encoding="utf-8") as f:
            data = json.load(f)
        for key, val in data.items():
            if isinstance(val, list):
                for e in val:
                    if isinstance(e, dict):
                        check(name + ":" + key, e)
    # the example_dataset texts are data files of the reference: compared byte for byte
    with open(os.path.join(HERE, "example_dataset.json"), encoding="utf-8") as f:
        files = json.load(f)["files"]
    for rel, text in files.items():
        checked += 1
        p = os.path.join(root, "example_dataset", rel)
        try:
            with open(p, encoding="utf-8") as f:
                if f.read() != text:
                    problems.append("example_dataset.json %s differs from the reference's file" % rel)
                else:
                    modes["exact"] += 1
        except OSError as e:
            problems.append("example_dataset.json %s: %s" % (rel, e))
    return problems, checked, modes


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if not os.path.isdir(root):
        print("reference checkout not found at %s: nothing verified" % root)
        return 0
    problems, checked, modes = verify(root)
    print("%d strings checked against the cited tests: %s" % (checked, modes))
    for p in problems:
        print("MISMATCH:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
