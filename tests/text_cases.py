"""shared by the CPU (oracle) and GPU (HIP) text / character(n) tests: a table with a text
column and a character(n) column inside heap tuples, and quals with the answers Python's own
bytes comparison gives (PostgreSQL's "C" collation is memcmp; character(n) ignores trailing
blanks -- opencl_textlib.h:150-399, codegen.c:616-629)"""
import numpy as np

from pg_strom_amd import kds

WORDS = [b"", b"a", b"ab", b"abc", b"abd", b"b", b"B", b"zebra", b"Zebra", b"hello world",
         b"hello", b"hello ", b"hello  ", b"\xc3\xa9t\xc3\xa9", b"\xff\x01", b"~", b"x" * 126, b"x" * 127,
         b"x" * 300, b"x" * 299 + b"y", b"MAIL", b"SHIP", b"TRUCK", b"AIR", b"REG AIR", b"RAIL", b"FOB"]


def text_table(n, seed, fmt="row", null_every=17):
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(WORDS), n)
    txt = [WORDS[i] for i in pick]
    # character(10): blank padded to 10 (longer words cut), the way PostgreSQL stores bpchar
    chr10 = [(WORDS[i][:10] + b" " * 10)[:10] for i in rng.integers(0, len(WORDS), n)]
    num = rng.integers(-50, 50, n).astype(np.int32)
    tnull = (np.arange(n) % null_every == 3)
    cols = [kds.Column("int4", num), kds.Column("text", txt, tnull),
            kds.Column("character", chr10), kds.Column("int8", np.arange(n, dtype=np.int64))]
    return kds.build_kds(fmt, cols), txt, chr10, num, tnull


def bpchar_key(b):
    return bytes(b).rstrip(b" ")


def expected_rows(qual_fn, txt, chr10, num, tnull):
    """rows (0-based) for which qual_fn(text or None, character, int) is True"""
    out = []
    for i in range(len(txt)):
        t = None if tnull[i] else txt[i]
        if qual_fn(t, chr10[i], int(num[i])) is True:
            out.append(i)
    return np.array(out, dtype=np.int32)


# (qual IR, python predicate over (text|None, character(10), int4), external params)
CASES = [
    ("(texteq (var 2 text) (const text 'hello'))",
     lambda t, c, n: None if t is None else t == b"hello", ()),
    ("(textne (var 2 text) (const text 'hello '))",
     lambda t, c, n: None if t is None else t != b"hello ", ()),
    ("(text_lt (var 2 text) (const text 'b'))",
     lambda t, c, n: None if t is None else t < b"b", ()),
    ("(text_ge (var 2 text) (param 0 text))",
     lambda t, c, n: None if t is None else t >= b"x" * 127, (b"x" * 127,)),
    ("(and (text_gt (var 2 text) (const text 'Zebra')) (text_le (var 2 text) (const text 'zebra')))",
     lambda t, c, n: None if t is None else (t > b"Zebra" and t <= b"zebra"), ()),
    ("(bpchareq (var 3 character) (const character 'hello'))",
     lambda t, c, n: bpchar_key(c) == b"hello", ()),
    ("(bpchareq (var 3 character) (param 0 character))",
     lambda t, c, n: bpchar_key(c) == b"REG AIR", (b"REG AIR   ",)),
    ("(or (bpcharlt (var 3 character) (const character 'B')) (bpcharge (var 3 character) (const character 'x')))",
     lambda t, c, n: bpchar_key(c) < b"B" or bpchar_key(c) >= b"x", ()),
    ("(and (int4gt (var 1 int4) (const int4 0)) (int4eq (bttextcmp (var 2 text) (const text 'abc')) (const int4 1)))",
     lambda t, c, n: (False if n <= 0 else None) if t is None else (n > 0 and t > b"abc"), ()),
    ("(int4le (bpcharcmp (var 3 character) (const character 'MAIL      ')) (const int4 0))",
     lambda t, c, n: bpchar_key(c) <= b"MAIL", ()),
    ("(isnull (var 2 text))", lambda t, c, n: t is None, ()),
    ("(case (when (isnull (var 2 text)) (const bool t)) (else (texteq (var 2 text) (const text ''))))",
     lambda t, c, n: True if t is None else t == b"", ()),
]
