"""Property tests (hypothesis) of the host I/O of the product library against the oracle's loaders -- themselves checked
value for value against the reference's graph_utilities.cc compiled as it lies (tests/test_oracle.py) -- and of the
renumbering / cache helpers.  CPU only: nothing here touches a GPU."""
import importlib
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle_lib as O

B = importlib.import_module("bipartitesbm-mcmc_amd")

# lines the reference's `stringstream >> a >> b` loop meets in the wild and in the weeds: numbers, signs, blanks of every
# kind, CRLF, empty lines, words, trailing junk, one number only, huge numbers
_token = st.one_of(st.integers(0, 99999).map(str), st.sampled_from(["", "x", "-3", "+7", "12abc", "1e3", "0007", "18446744073709551615"]))
_sep = st.sampled_from([" ", "\t", "  ", " \t ", ","])
_line = st.builds(lambda lead, a, s1, b, s2, c, cr: lead + a + s1 + b + s2 + c + cr,
                  st.sampled_from(["", " ", "\t"]), _token, _sep, _token, st.sampled_from(["", " ", "\t"]),
                  st.sampled_from(["", "junk", "5"]), st.sampled_from(["", "\r"]))
_text = st.lists(_line, min_size=0, max_size=30).map(lambda ls: "\n".join(ls)) .flatmap(
    lambda t: st.sampled_from([t, t + "\n", t + "\n\n"]))


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=_text)
def test_edge_list_and_membership_scanners_equal_the_oracles(tmp_path, text):
    p = tmp_path / "f.txt"
    p.write_bytes(text.encode())
    a, b = B.load_edge_list(str(p))
    oa, ob = O.load_edge_list(str(p))
    assert (a == oa).all() and (b == ob).all() and len(a) == len(oa)
    m = B.load_memberships(str(p))
    om = O.load_memberships(str(p))
    assert (m == om).all() and len(m) == len(om)


_graph = st.integers(1, 40).flatmap(lambda na: st.integers(1, 40).flatmap(lambda nb: st.lists(
    st.tuples(st.integers(0, na - 1), st.integers(na, na + nb - 1)), min_size=0, max_size=200).map(lambda e: (na, nb, e))))


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(g=_graph)
def test_csr_cache_and_renumbering_on_random_graphs(tmp_path, g):
    na, nb, edges = g
    n = na + nb
    p = tmp_path / "g.el"
    p.write_text("".join("%d\t%d\n" % e for e in edges))
    want = O.edge_to_csr(np.array([e[0] for e in edges], dtype=np.uint64), np.array([e[1] for e in edges], dtype=np.uint64), n)
    for _ in range(2):  # build the cache, then read it
        rp, cl = B.load_graph(str(p), n, cache=True)
        assert (rp == want[0]).all() and (cl == want[1]).all()
    assert B.load_graph.last_cache_hit
    os.remove(str(p) + ".bisbm_csr")
    lo = B.locality_order(rp, cl, na, nb)
    assert sorted(lo.new_id[:na]) == list(range(na)) and sorted(lo.new_id[na:]) == list(range(na, n))
    rp2, cl2 = lo.apply(rp, cl)
    # same multigraph under the renumbering, rows in the same edge order
    for v in range(n):
        assert list(cl2[rp2[lo.new_id[v]]:rp2[lo.new_id[v] + 1]]) == [lo.new_id[u] for u in cl[rp[v]:rp[v + 1]]]
