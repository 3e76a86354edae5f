"""svjg/filter.py: host_line — what the host does with a line the kernels set aside (SVJG_EXC_ASK_HOST): Python's own int() / float() decide, and
numbers the kernels cannot hold (more than 18 digits) are rewritten within range WITH THE SAME MEANING for the two comparisons they enter
(filter-alignments.py:260-271: left_sum - Ts >= d_over and right_sum - (Tlen - Te - 1) >= d_over).  Property test against Python's big integers."""
import random
import sys

import pytest

from svjg import filter as flt

COLS = ["r", "100", "0", "100", "+", ">1:1-500>1:501-900", "900", "5", "890", "90", "100", "60", "tp:A:P"]


def _line(**kw):
    c = list(COLS)
    for k, v in kw.items():
        c[int(k[1:])] = str(v)
    return "\t".join(c) + "\n"


def _outcome(cols, left, right, d_over=100):
    tlen, ts, te = int(cols[6]), int(cols[7]), int(cols[8])
    return left - ts >= d_over, right - (tlen - te - 1) >= d_over


def test_rewritten_columns_mean_the_same():
    rng = random.Random(6)
    big = [10 ** 18, 10 ** 18 + 1, 10 ** 19, 2 ** 61, 2 ** 63, 2 ** 64, 10 ** 30, 10 ** 300, 10 ** 4299]
    for _ in range(4000):
        base = rng.choice(big) * rng.choice((1, 1, -1))
        tlen = base + rng.randrange(-2000, 2000) if rng.random() < 0.7 else rng.randrange(0, 5000)
        te = base + rng.randrange(-2000, 2000) if rng.random() < 0.7 else rng.randrange(0, 5000)
        ts = rng.choice((rng.randrange(0, 3000), base, -base, base + rng.randrange(-500, 500)))
        if max(abs(tlen), abs(ts), abs(te)) <= flt._COL_CAP:
            continue
        text = _line(c6=tlen, c7=ts, c8=te)
        out = flt.host_line(text).decode().rstrip("\n").split("\t")
        assert all(len(out[i].lstrip("+-")) <= 18 for i in (6, 7, 8)), out[6:9]           # within the kernels' range
        for _ in range(20):                                                                # any path sums the kernels can hold
            left, right = rng.randrange(-10 ** 16, 10 ** 17), rng.randrange(-10 ** 16, 10 ** 17)
            if rng.random() < 0.3:
                right = (tlen - te - 1) + rng.randrange(98, 103) if abs(tlen - te - 1) < 10 ** 17 else right
                left = ts + rng.randrange(98, 103) if abs(ts) < 10 ** 17 else left
            assert _outcome(text.rstrip("\n").split("\t"), left, right) == _outcome(out, left, right), (tlen, ts, te, left, right, out[6:9])
        assert out[:6] == COLS[:6] and out[9:] == COLS[9:]                                  # nothing else moves


def test_the_other_columns_and_what_python_refuses():
    out = flt.host_line(_line(c1=10 ** 40, c9=-10 ** 25, c11="1_0")).decode().split("\t")
    assert out[1] == "1" and out[9] == "1" and out[11] == "1_0"                            # they only have to BE integers (filter-alignments.py:185-191)
    assert flt.host_line(_line(c10=10 ** 40)).decode().split("\t")[10] == "1"             # Alen: only "zero or not" matters (:196)
    with pytest.raises(ZeroDivisionError):
        flt.host_line(_line(c10="0" * 30))
    out = flt.host_line(_line(c6="0" * 30 + "900", c7="0_0_0_0_0_0_0_0_0_0_0_0_0_0_0_0_0_0_0_5", c1="+" + "0" * 40)).decode().split("\t")
    assert out[6] == "900" and out[7] == "5" and out[1] == "0"                             # more digits than the value needs: written as the value
    with pytest.raises(OverflowError):                                                     # Am / Alen beyond a double (:196)
        flt.host_line(_line(c9=10 ** 400))
    assert flt.host_line(_line(c9=10 ** 400).rstrip("\n") + "\tid:f:0.9\n")                # with the tag the quotient is never formed
    if hasattr(sys, "get_int_max_str_digits") and sys.get_int_max_str_digits():
        lim = sys.get_int_max_str_digits()
        flt.host_line(_line(c6="1" * lim))                                                 # the interpreter's own limit decides, as for the reference
        with pytest.raises(ValueError):
            flt.host_line(_line(c6="1" * (lim + 1)))
    with pytest.raises(ValueError):
        flt.host_line(_line(c7="5\x1f"))                                                   # int() does not strip what rstrip() strips
    assert flt.host_line(_line(c7=" 5 ").rstrip("\n") + "\x1f\n").decode().split("\t")[7] == " 5 "


def test_unreadable_node_names():
    """a line set aside twice met a node name the exact routine cannot read; where Python's arithmetic on every such name dies with one exception class,
    that is what the reference dies with — else (a number Python computes with) the line is refused"""
    def line(path):
        c = list(COLS); c[5] = path
        return ("\t".join(c) + "\n").encode("utf-8")
    assert isinstance(flt._name_error(line(">1:1-500>1:1 9051-20000")), ValueError)        # a blank INSIDE a number: int() dies
    assert isinstance(flt._name_error(line(">1:1-500>1:5é-9>1:xé-9")), ValueError)
    assert isinstance(flt._name_error(line("<1:é7>1:1-500")), IndexError)                  # no '-': coords.split("-")[1]
    assert flt._name_error(line(">1:1-500>1:٣-9")) is None                                # Unicode digits: Python computes 9 - 3 + 1
    assert flt._name_error(line(">1:1-500>1:1- 37500")) is None                           # a Unicode blank AROUND a number: int() strips it
    assert flt._name_error(line(">1:1-500>1:1-99999999999999999999")) is None                  # twenty digits: Python computes
    assert flt._name_error(line(">1:1-500>1:5é-9>1:é7")) is None                     # ValueError or IndexError, depending on which the reference meets: refused
    assert flt._name_error(line(">1:1-500>1:501-900")) is None                                 # nothing unreadable: not this function's case
