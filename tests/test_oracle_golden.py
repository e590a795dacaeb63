"""CPU: the oracle (oracle/gdsp_oracle.c) against the golden vectors recorded from the
unmodified reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from backends import OracleBackend
from conftest import bits_equal, first_diff, golden
from pipeline import Runner

# ops whose reference result the oracle reproduces bit for bit on every fixture
CASES = golden().vector_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_vectors(case, gold):
    chroms = [tuple(c) for c in case["chroms"]]
    r = Runner(OracleBackend(), chroms, gold.inputs(case), case.get("files")).run(case["pipeline"])
    if "percentile" not in case["pipeline"] or "--preserve" in case["pipeline"]:   # else the reference leaves the signal scrambled
        want = gold.outputs(case)
        for c, _ in chroms:
            got = r.result(c)
            assert bits_equal(got, want[c]), "%s %s first differing index %s" % (
                case["name"], c, first_diff(got, want[c]))
    for name, hexval in case["globals"].items():
        assert hexval is not None, "reference did not set " + name
        assert name in r.globals, name
        assert float.fromhex(hexval) == r.globals[name], (name, float.fromhex(hexval), r.globals[name])


def test_hann_taps_match_reference_impulse_response(gold):
    from oracle import cpu
    for W in (3, 5, 11, 101, 1001):
        case = gold.cases["hann_impulse_W%d" % W]
        out = gold.outputs(case)["chrA"]
        n = out.size
        taps = cpu.hann_window(W)
        centre = n // 2
        # response to a unit impulse at `centre` is the tap vector reversed (symmetric) around it
        got = out[centre - (W - 1) // 2: centre + (W - 1) // 2 + 1]
        assert bits_equal(got, taps[::-1]), W
        assert bits_equal(taps, taps[::-1])


def test_known_answers_from_survey(gold):
    """W=101 centre / outermost / second taps at 20 decimals."""
    from oracle import cpu
    t = cpu.hann_window(101)
    assert "%.20f" % t[50] == "0.01960784313725489822"
    assert "%.20f" % t[0] == "0.00001859481630348917"
    assert "%.20f" % t[1] == "0.00007430872870651239"
    t5 = cpu.hann_window(5)
    assert ["%.20f" % x for x in t5[:3]] == ["0.08333333333333332871", "0.25000000000000000000",
                                              "0.33333333333333337034"]
