"""Pins tests/fp64_definition.py (the fp64 statement of the RM2 definition used for the ALL-ROWS precision measurement at
full size) to what is pinned already: the reference's 507 golden triples and the brute-force oracle (the reference's own
loop nest, oracle/rm2_oracle.c).  Runs on the CPU (torch fp64)."""
import numpy as np
import torch

import oracle
from fp64_definition import compare_with_definition, fp64_scores
from test_oracle_golden import run_golden
from util import synth


def _rows(r):
    return {"user": r["rec_user"], "item": r["rec_item"], "score": r["rec_score"], "cluster": r["rec_cluster"]}


def test_definition_reproduces_the_references_507_triples(rm_golden):
    exp = np.asarray(rm_golden["recommendations"])
    rows = {"user": exp[:, 0].astype(np.int32), "item": exp[:, 1].astype(np.int32), "score": exp[:, 2].astype(np.float32)}
    trip = tuple(torch.as_tensor(a) for a in rm_golden["coo"])
    ref = fp64_scores(trip, rows, rm_golden["params"]["lambda"], rm_golden["numberOfItems"],
                      clustering=(rm_golden["map_user"], rm_golden["map_cluster"]), device="cpu")
    assert np.max(np.abs(ref - exp[:, 2])) <= 1e-4                    # the reference's own criterion (HadoopIntegrationTest.java:53)
    rep = compare_with_definition(rows, ref)
    assert rep["worst"] <= 2e-7 and rep["n_over"] == 0                # the residue is the fixture's 6-decimal print


def test_definition_equals_the_brute_force_oracle_on_clustered_half_star_data():
    S = synth()
    u, i, s, facts = S.generate("tiny")
    uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
    cl = S.hash_clustering(uu, 4)
    cl[5] = 0
    keep = uu != 7                                                     # user 7 is unmapped -> cluster 0 (quirk Q2)
    for lam in (0.1, 0.5, 0.9):
        r = oracle.rm2(u.numpy(), i.numpy(), s.numpy(), lam=lam, number_of_items=facts["n_items"], number_of_recommendations=25,
                       number_of_clusters=4, map_user=uu[keep], map_cluster=cl[keep], n_threads=4)
        ref = fp64_scores((u, i, s), _rows(r), lam, facts["n_items"], clustering=(uu[keep], cl[keep]), device="cpu",
                          elem_budget=5000, col_chunk=64)             # small budgets: every batching path is taken
        rep = compare_with_definition(_rows(r), ref)
        assert rep["rows"] == len(r["rec_user"]) > 4000
        assert rep["worst"] <= 1.5e-7, rep                             # float32 cast of the oracle's rows
        # and in fp64, against the oracle's Gram variant too
        g = oracle.rm2_gram(u.numpy(), i.numpy(), s.numpy(), lam=lam, number_of_items=facts["n_items"], number_of_recommendations=25,
                            number_of_clusters=4, map_user=uu[keep], map_cluster=cl[keep], n_threads=4)
        np.testing.assert_array_equal(g["rec_item"], r["rec_item"])
