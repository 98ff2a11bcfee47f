#!/usr/bin/env python3
"""
Regenerates the reference's regression fixture table `gpupreagg_test`
(input/sql/agg_init.sql:43-108) WITHOUT PostgreSQL, and extracts the
expected results of the reference's own regression suites
(expected/{nogrp,group,where,zero}_agg.out -- produced by stock PostgreSQL
with the extension disabled, input/make_expected.sh:22-28) into JSON.

Run in the build container (reads /root/reference/expected/*.out as data):
    python tests/golden/make_gpupreagg_fixture.py
Writes tests/golden/gpupreagg_test.npz and tests/golden/expected_agg.json.

How the table is reproduced: PostgreSQL 9.4 `setseed(0)` is
srandom((unsigned)(0 * MAX_RANDOM_VALUE)) and `random()` is
(double)random() / 2^31 on glibc, whose random() (TYPE_3 additive feedback,
r[i] = r[i-3] + r[i-31]) is restated below.  Target-list entries are
evaluated left to right per row; CASE evaluates lazily; and because the
INSERTs put set-returning generate_series() calls in the target list,
ExecTargetList evaluates every non-SRF expression one extra time when the
series end -- so each INSERT consumes 10001 rows' worth of random() calls.
The generator is verified below against known answers of the reference's
expected output (count/sum/min/max of several columns) before anything is
written.
"""
import json
import os
import re
import sys
from decimal import Decimal, ROUND_HALF_UP

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class GlibcRandom(object):
    """glibc random_r(), TYPE_3 (degree 31, separation 3)"""

    def __init__(self, seed):
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1], 127773)
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word
        for i in range(31, 34):
            r[i] = r[i - 31]
        self.r = r
        for _ in range(310):
            self._next()

    def _next(self):
        r = self.r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        if len(r) > 4096:
            del r[:len(r) - 64]
        return v >> 1

    def random(self):
        """SQL random(): [0,1)"""
        return self._next() / 2147483648.0


def float8_to_numeric(x):
    """float8_numeric(): sprintf("%.*g", DBL_DIG, val) then numeric_in"""
    return Decimal("%.15g" % x)


def numeric_round(d, scale):
    return d.quantize(Decimal(1).scaleb(-scale), rounding=ROUND_HALF_UP)


def rint(x):
    return int(np.rint(x))


def generate():
    rng = GlibcRandom(0)
    rnd = rng.random
    cols = {k: [] for k in ("id", "key", "smlint_x", "integer_x", "bigint_x", "real_x", "float_x",
                            "nume_x", "smlsrl_x", "serial_x", "bigsrl_x")}

    def block(first_id, first_key, f):
        # 10000 stored rows + one discarded evaluation (see module docstring)
        for i in range(10001):
            row = {}
            row["id"] = first_id + i
            row["key"] = first_key + (i % 10)
            for name, scale in (("smlint_x", 32767), ("integer_x", 2147483647),
                                ("bigint_x", 9223372036854775807)):
                row[name] = None if rnd() > 0.95 else rint(f(rnd()) * scale / 1000)
            row["real_x"] = None if rnd() > 0.95 else numeric_round(float8_to_numeric(f(rnd())), 4)
            row["float_x"] = None if rnd() > 0.95 else numeric_round(float8_to_numeric(f(rnd())), 13)
            row["nume_x"] = None if rnd() > 0.95 else float8_to_numeric(f(rnd()))
            row["smlsrl_x"] = rint(f(rnd()) * 32767 / 1000)
            row["serial_x"] = rint(f(rnd()) * 2147483647 / 1000)
            row["bigsrl_x"] = rint(f(rnd()) * 9223372036854775807 / 1000)
            if i < 10000:
                for k in cols:
                    cols[k].append(row[k])

    block(1, 1, lambda x: x)
    block(10001, 11, lambda x: x * -1)
    block(20001, 21, lambda x: x * 2 - 1)
    for i in range(10000):
        cols["id"].append(30001 + i)
        for k in ("key", "smlint_x", "integer_x", "bigint_x", "real_x", "float_x", "nume_x"):
            cols[k].append(None)
        for k in ("smlsrl_x", "serial_x", "bigsrl_x"):
            cols[k].append(0)
    return cols


def generate_overflow():
    """gpupreagg_overflow_test (agg_init.sql:122-200): the same seed, columns pinned at the
    edges of their types -- 32767 / 2147483647 / 9223372036854775807 / 1e38 / 1e308 and
    21-digit numerics -- so that device partial sums overflow and chunks go back to the CPU"""
    rng = GlibcRandom(0)
    rnd = rng.random
    cols = {k: [] for k in ("id", "key", "smlint_x", "integer_x", "bigint_x", "real_x", "float_x",
                            "nume_x", "smlsrl_x", "serial_x", "bigsrl_x")}
    NUM21 = 1000000000000000000000.0          # the literal is numeric, the product float8

    def block(first_id, first_key, mode):
        sign = {"pos": 1, "neg": -1}.get(mode)
        for i in range(10001):                 # 10000 stored rows + one discarded evaluation
            row = {"id": first_id + i, "key": first_key + (i % 10)}
            if mode in ("pos", "neg"):
                row["smlint_x"] = None if rnd() > 0.95 else (32767 if sign > 0 else -32768)
                row["integer_x"] = None if rnd() > 0.95 else (2147483647 if sign > 0 else -2147483648)
                row["bigint_x"] = None if rnd() > 0.95 else (9223372036854775807 if sign > 0
                                                             else -9223372036854775808)
                row["real_x"] = None if rnd() > 0.95 else sign * 1.0e38
                row["float_x"] = None if rnd() > 0.95 else sign * 1.0e308
                if rnd() > 0.95:
                    row["nume_x"] = None
                else:
                    v = float8_to_numeric(np.floor(rnd() * NUM21))
                    row["nume_x"] = v if sign > 0 else v * -1
                row["smlsrl_x"] = rint(rnd() * (32767 * sign))
                row["serial_x"] = rint(rnd() * (2147483647 * sign))
                row["bigsrl_x"] = rint(rnd() * (9223372036854775807 * sign))
            else:
                row["smlint_x"] = None if rnd() > 0.95 else rint((rnd() * 2 - 1) * 32767)
                row["integer_x"] = None if rnd() > 0.95 else rint((rnd() * 2 - 1) * 2147483647)
                row["bigint_x"] = None if rnd() > 0.95 else rint((rnd() * 2 - 1) * 9223372036854775807)
                row["real_x"] = None if rnd() > 0.95 else (rnd() * 2 - 1) * 1.0e38
                row["float_x"] = None if rnd() > 0.95 else (rnd() * 2 - 1) * 1.0e308
                row["nume_x"] = None if rnd() > 0.95 else float8_to_numeric(np.floor((rnd() * 2 - 1) * NUM21))
                row["smlsrl_x"] = rint((rnd() * 2 - 1) * 32767)
                row["serial_x"] = rint((rnd() * 2 - 1) * 2147483647)
                row["bigsrl_x"] = rint((rnd() * 2 - 1) * 9223372036854775807)
            if i < 10000:
                for k in cols:
                    cols[k].append(row[k])

    block(1, 1, "pos")
    block(10001, 11, "neg")
    block(20001, 21, "mix")
    for i in range(10000):
        cols["id"].append(30001 + i)
        for k in ("key", "smlint_x", "integer_x", "bigint_x", "real_x", "float_x", "nume_x"):
            cols[k].append(None)
        for k in ("smlsrl_x", "serial_x", "bigsrl_x"):
            cols[k].append(0)
    return cols


def overflow_to_arrays(cols):
    out = {}

    def intcol(name, dtype):
        out[name] = np.array([0 if v is None else v for v in cols[name]], dtype=dtype)
        out[name + "_isnull"] = np.array([v is None for v in cols[name]], dtype=np.uint8)

    for name, dt in (("id", np.int32), ("key", np.int32), ("smlint_x", np.int16), ("integer_x", np.int32),
                     ("bigint_x", np.int64), ("smlsrl_x", np.int16), ("serial_x", np.int32), ("bigsrl_x", np.int64)):
        intcol(name, dt)
    out["real_x"] = np.array([0.0 if v is None else np.float32(v) for v in cols["real_x"]], dtype=np.float32)
    out["real_x_isnull"] = np.array([v is None for v in cols["real_x"]], dtype=np.uint8)
    out["float_x"] = np.array([0.0 if v is None else v for v in cols["float_x"]], dtype=np.float64)
    out["float_x_isnull"] = np.array([v is None for v in cols["float_x"]], dtype=np.uint8)
    out["nume_x"] = np.array(["" if v is None else format(v, "f") for v in cols["nume_x"]])
    out["nume_x_isnull"] = np.array([v is None for v in cols["nume_x"]], dtype=np.uint8)
    assert len(out["id"]) == 40000
    return out


def verify_overflow(arr, expected):
    """known answers of expected/overflow_agg.out (stock PostgreSQL) for the regenerated table"""
    ok = True
    key, keyn = arr["key"], arr["key_isnull"]
    for q in expected:
        m = re.match(r"select key, (count|sum)\((smlint_x|integer_x|bigint_x|nume_x|serial_x)\)::", q["sql"])
        if not m or q.get("error"):
            continue
        func, col = m.group(1), m.group(2)
        for row in q["rows"]:
            if row[0] == "":
                continue
            sel = (key == int(row[0])) & (keyn == 0) & (arr[col + "_isnull"] == 0)
            if func == "count":
                got = int(sel.sum())
            elif col == "nume_x":
                got = sum((Decimal(x) for x in arr[col][sel]), Decimal(0))
            else:
                got = sum(int(x) for x in arr[col][sel])
            good = (str(got) == row[1])
            ok &= good
            if not good:
                print("  MISMATCH %s(%s) key %s: %s expected %s" % (func, col, row[0], got, row[1]))
        print("  %-5s(%-9s) per key: %s" % (func, col, "ok" if ok else "MISMATCH"))
    return ok


def to_arrays(cols):
    out = {}
    n = len(cols["id"])

    def intcol(name, dtype):
        isnull = np.array([v is None for v in cols[name]], dtype=np.uint8)
        vals = np.array([0 if v is None else v for v in cols[name]], dtype=dtype)
        out[name] = vals
        out[name + "_isnull"] = isnull

    intcol("id", np.int32)
    intcol("key", np.int32)
    intcol("smlint_x", np.int16)
    intcol("integer_x", np.int32)
    intcol("bigint_x", np.int64)
    intcol("smlsrl_x", np.int16)
    intcol("serial_x", np.int32)
    intcol("bigsrl_x", np.int64)
    # real: numeric -> float4 goes through text -> strtod -> (float4)
    out["real_x"] = np.array([0.0 if v is None else np.float32(float(str(v))) for v in cols["real_x"]],
                             dtype=np.float32)
    out["real_x_isnull"] = np.array([v is None for v in cols["real_x"]], dtype=np.uint8)
    out["float_x"] = np.array([0.0 if v is None else float(str(v)) for v in cols["float_x"]],
                              dtype=np.float64)
    out["float_x_isnull"] = np.array([v is None for v in cols["float_x"]], dtype=np.uint8)
    out["nume_x"] = np.array(["" if v is None else str(v) for v in cols["nume_x"]])
    out["nume_x_isnull"] = np.array([v is None for v in cols["nume_x"]], dtype=np.uint8)
    assert len(out["id"]) == n == 40000
    return out


def verify(arr):
    """known answers from expected/nogrp_agg.out (stock PostgreSQL)"""
    def col(name):
        return arr[name][arr[name + "_isnull"] == 0]
    checks = [
        ("count(smlint_x)", len(col("smlint_x")), 28477),
        ("sum(smlint_x)", int(col("smlint_x").astype(np.int64).sum()), 289),
        ("max(smlint_x)", int(col("smlint_x").max()), 33),
        ("count(integer_x)", len(col("integer_x")), 28511),
        ("sum(integer_x)", int(col("integer_x").astype(np.int64).sum()), 99027633),
        ("max(integer_x)", int(col("integer_x").max()), 2147112),
        ("min(integer_x)", int(col("integer_x").min()), -2147350),
        ("sum(bigint_x)", int(col("bigint_x").sum()), -55757751021379520),
    ]
    ok = True
    for name, got, want in checks:
        flag = "ok" if got == want else "MISMATCH"
        ok &= got == want
        print("  %-18s %22d  expected %22d  %s" % (name, got, want, flag))
    fs = float(np.sum(col("float_x")))
    print("  sum(float_x) %.10f expected 31.7956865663" % fs)
    ok &= abs(fs - 31.7956865663) < 1e-9
    return ok


def parse_out(path):
    """psql regression output -> [{sql, columns, rows}] (+ "error": text for a query that
    ended in ERROR, + "notices": [...] for NOTICE lines printed in front of the result)"""
    lines = open(path).read().split("\n")
    res = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.lower().startswith("select"):
            sql = ln.strip()
            notices = []
            while i + 1 < len(lines) and lines[i + 1].startswith("NOTICE:"):
                notices.append(lines[i + 1][len("NOTICE:"):].strip())
                i += 1
            if i + 1 < len(lines) and lines[i + 1].startswith("ERROR:"):
                res.append({"sql": re.sub(r"\s+", " ", sql), "columns": [], "rows": [],
                            "error": lines[i + 1][len("ERROR:"):].strip()})
                i += 2
                continue
            header = lines[i + 1]
            sep = lines[i + 2] if i + 2 < len(lines) else ""
            if not re.match(r"^-+(\+-+)*$", sep):
                i += 1
                continue
            columns = [c.strip() for c in header.split("|")]
            rows = []
            j = i + 3
            while j < len(lines) and not re.match(r"^\(\d+ rows?\)$", lines[j]):
                rows.append([c.strip() for c in lines[j].split("|")])
                j += 1
            rec = {"sql": re.sub(r"\s+", " ", sql), "columns": columns, "rows": rows}
            if notices:
                rec["notices"] = notices
            res.append(rec)
            i = j
        i += 1
    return res


def main():
    print("regenerating gpupreagg_test ...")
    arr = to_arrays(generate())
    if not verify(arr):
        print("generator does not reproduce the reference's fixture", file=sys.stderr)
        sys.exit(1)
    np.savez_compressed(os.path.join(HERE, "gpupreagg_test.npz"), **arr)
    expected = {}
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg", "recheck_agg", "overflow_agg"):
        expected[suite] = parse_out(os.path.join(REF, "expected", suite + ".out"))
        print("  %s: %d queries" % (suite, len(expected[suite])))
    print("regenerating gpupreagg_overflow_test ...")
    oarr = overflow_to_arrays(generate_overflow())
    if not verify_overflow(oarr, expected["overflow_agg"]):
        print("generator does not reproduce the reference's overflow fixture", file=sys.stderr)
        sys.exit(1)
    np.savez_compressed(os.path.join(HERE, "gpupreagg_overflow_test.npz"), **oarr)
    with open(os.path.join(HERE, "expected_agg.json"), "w") as fp:
        json.dump(expected, fp, indent=0)
    print("written")


if __name__ == "__main__":
    main()
