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


# Words in many scripts plus characters at the edges of the ranges that the GPU classifier settles from
# bit planes (letters next to symbols of the same block, unassigned code points, pictographs with and
# without the Alphabetic property, every White_Space character).
SCRIPT_WORDS = ["прибыль", "Ёж", "ҐґҒ", "\u0482\u0483", "Ԁԯ", "\u0530", "Աբգ", "Ֆ\u0557ՙ", "×", "÷", "ÀÖØöøÿ", "ĀſƀɏɐʯʰˁDŽ",
                "\u02c2\u02c5ˆ", "ΆΈΊΌΎΡ\u03a2Σ", "ϵ϶Ϸ", "\u0378\u0384·", "שלום", "\u05be\u05c0", "سلام", "٠١٢", "ۿ۽",
                "हिन्दी", "ไทย", "ひらがな", "カタカナ", "・ー", "\u3040\u3097", "㐀䶿", "䷀䷿", "一鿿", "ꀀ꒍", "가힣", "\ud7a4\ud7b0ퟻ",
                "한국어", "🄰🅉🅰", "🈀🌀🚀🫸", "\U0001fbf0", "𐀀𐌀", "𠀀𪛟", "\U0002fa1d\U0003134a", "—–‘’“”•…", "€₠⃀", "ⁱⁿ₀ₐ", "ℂ℃ℊ",
                "a\u00a0b", "c\u1680d", "e\u2000f\u200ag", "h\u2028i\u2029j", "k\u202fl\u205fm", "n\u3000o", "p\u0085q",
                "\u200b\u200e", "\ufeffBOM", "ｆｕｌｌ", "ﬁﬂ", "𝐀𝑏𝒸"]


def script_corpus(seed, n_records):
    """Records whose lines mix SCRIPT_WORDS with numbers so that all three gates see them."""
    rng = random.Random(seed)
    recs = []
    for _ in range(n_records):
        lines = []
        for _ in range(rng.randint(1, 8)):
            toks = []
            for _ in range(rng.randint(1, 16)):
                k = rng.random()
                toks.append(rng.choice(SCRIPT_WORDS) if k < 0.6 else rng.choice(NUMS) if k < 0.9 else rng.choice(WORDS))
            line = rng.choice(["", " ", "| "]) + rng.choice([" ", " ", " | ", "\t"]).join(toks) + rng.choice(["", " |", " "])
            lines.append(line)
        recs.append(rng.choice(EOLS[:5]).join(lines).encode("utf-8"))
    return recs
