"""Seeded generators of adversarial text for parity tests (small sizes; pure Python)."""
import random

WORDS = ["alpha", "beta", "Revenue", "NOVEX", "INDUSTRIES", "grew", "the", "a", "of", "Société",
         "Ünternehmen", "北京工业有限公司", "東京", "α", "β", "إيرادات", "הכנסות", "Springfield", "Corp.",
         "Q3", "3rd", "FY2025", "x", "I"]
NUMS = ["42", "524", "10788", "143", "1995", "10,788.0", "14.3", "(0.8)", "+12%", "-8%", "$4.2B",
        "€100", "¥500", "₹200", "99999", "7", "0"]
SYMS = ["—", "&", "🚀", "🌏", "€", "--", ":", "|", "||", "-", "***", "."]
WS = [" ", " ", " ", " ", "  ", "\t", " ", "　", " ", "\u0085", "\x0b", "\x0c", " "]
EOLS = ["\n", "\n", "\n", "\r\n", "\r", "\n\n", "\r\r\n", "\n\r"]


def _line(rng):
    kind = rng.random()
    if kind < 0.25:      # prose
        toks = [rng.choice(WORDS) if rng.random() < 0.9 else rng.choice(NUMS)
                for _ in range(rng.randint(1, 14))]
    elif kind < 0.5:     # OCR row
        toks = [rng.choice(NUMS) for _ in range(rng.randint(0, 3))]
        toks += [rng.choice(WORDS) for _ in range(rng.randint(1, 4))]
        toks += [rng.choice(NUMS + SYMS[:5]) for _ in range(rng.randint(2, 14))]
        if rng.random() < 0.3:
            rng.shuffle(toks)
    elif kind < 0.6:     # numeric / symbol only
        toks = [rng.choice(NUMS + SYMS) for _ in range(rng.randint(0, 8))]
    elif kind < 0.8:     # table row
        ncell = rng.randint(0, 5)
        cells = []
        for _ in range(ncell):
            c = rng.random()
            if c < 0.2:
                cells.append(rng.choice(["", " ", "---", ":--:", "  "]))
            else:
                cells.append(rng.choice(WS[:6]) * rng.randint(0, 2) +
                             rng.choice(WS[:4]).join(rng.choice(WORDS + NUMS)
                                                     for _ in range(rng.randint(1, 3))) +
                             rng.choice(WS[:6]) * rng.randint(0, 2))
        row = "|".join(cells)
        if rng.random() < 0.8:
            row = "|" + row
        if rng.random() < 0.8:
            row = row + "|"
        return rng.choice(["", " ", "  ", "\t"]) + row + rng.choice(["", " ", " "])
    elif kind < 0.85:    # blank / whitespace only
        return rng.choice(["", " ", "\t ", "　"])
    else:                # mixed with odd spacing
        toks = [rng.choice(WORDS + NUMS + SYMS) for _ in range(rng.randint(1, 20))]
    out = rng.choice(["", "", " ", "  "])
    for i, t in enumerate(toks):
        if i:
            out += rng.choice(WS)
        out += t
    out += rng.choice(["", "", " ", "\t"])
    return out


def record(rng, max_lines=12):
    n = rng.choice([0, 1, 1, 2, 3, 5, max_lines])
    s = ""
    for i in range(n):
        s += _line(rng)
        if i + 1 < n or rng.random() < 0.5:
            s += rng.choice(EOLS)
    if rng.random() < 0.05:
        s = ""
    return s.encode("utf-8")


def long_line_record(rng, ntok):
    """one long line (as the File/CSV sources produce after whitespace collapsing)"""
    toks = [rng.choice(WORDS) if rng.random() < 0.6 else rng.choice(NUMS + SYMS[:4])
            for _ in range(ntok)]
    return " ".join(toks).encode("utf-8")


def digit_heavy_long_line(rng, ntok):
    toks = [rng.choice(WORDS) if rng.random() < 0.12 else rng.choice(NUMS + SYMS[:4])
            for _ in range(ntok)]
    return " ".join(toks).encode("utf-8")


def corpus(seed, n_records, max_lines=12):
    rng = random.Random(seed)
    recs = []
    for i in range(n_records):
        r = rng.random()
        if r < 0.03:
            recs.append(long_line_record(rng, rng.randint(50, 3000)))
        elif r < 0.05:
            recs.append(digit_heavy_long_line(rng, rng.randint(20, 1500)))
        else:
            recs.append(record(rng, max_lines))
    return recs
