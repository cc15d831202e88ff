"""Config C4 of BASELINE.json: Netflix-Prize-shaped synthetic ratings (480 189 x 17 770, 100 M ratings, integers 1..5),
RM2 top-100 + item-item similarity.  The only config that takes the two-chunk seed of the branch and bound (N = 100) and
the single-LDS-chunk row kernel.  Same checks as at ML-25M shape (tests/fullsize_checks.py), including the all-rows
comparison of the pruned job with the plain full pass (48 M rows)."""
import pytest

from fullsize_checks import all_rows_against_fp64_definition, assert_same_lists, check_itemsim, compare_itemsim_builds, check_rm2, load_shape, run_rm2

pytestmark = pytest.mark.gpu
LAM, TOPN = 0.1, 100


@pytest.fixture(scope="module")
def data():
    return load_shape("netflix")


@pytest.fixture(scope="module")
def pruned(data):
    return run_rm2(data, TOPN, LAM)


def test_rm2_netflix_shape(data, pruned):
    rows, sums, st = pruned
    assert st["blocks_total"] > 0 and st["blocks_survived"] < 0.05 * st["blocks_total"]
    worst = check_rm2(data, rows, sums, st, TOPN, LAM, n_picks=6)
    print("netflix-shape worst relative error vs fp64 definition: %.2e" % worst)


def test_rm2_netflix_all_rows_against_the_fp64_definition(data, pruned):
    """all 48 M rows of the top-100 job against the definition in fp64 (tests/fp64_definition.py)"""
    all_rows_against_fp64_definition(data, pruned[0], LAM, 1, "Netflix shape, one cluster")


def test_rm2_netflix_pruned_equals_full_pass_all_rows(data, pruned):
    rows, _, st = pruned
    rows_full, _, st_full = run_rm2(data, TOPN, LAM, env={"FY_PRUNE": "0"})
    assert st_full["blocks_total"] == 0 and st_full["recs"] == st["recs"]
    n_diff, worst = assert_same_lists(rows, rows_full, score_rtol=1e-5)
    print("pruned vs full pass: %d rows, %d differ (ties at a cut-off), worst score difference %.2e" % (len(rows["user"]), n_diff, worst))
    assert n_diff <= 1e-5 * len(rows["user"])


def test_rm2_netflix_panel_mode_50_clusters_equals_full_pass_all_rows(data):
    """50 hashed clusters (9 600 users each) in column-panel mode against the same job with FY_PRUNE=0: all 48 M rows."""
    rows, _, st = run_rm2(data, TOPN, LAM, clusters=50)
    assert st["panel_clusters"] == 50 and st["blocks_total"] > 0
    rows_full, _, st_full = run_rm2(data, TOPN, LAM, env={"FY_PRUNE": "0"}, clusters=50)
    assert st_full["panel_clusters"] == 0 and st_full["blocks_total"] == 0 and st_full["recs"] == st["recs"]
    n_diff, worst = assert_same_lists(rows, rows_full, score_rtol=1e-5)
    print("panel mode vs full pass, 50 clusters: %d rows, %d differ, worst score difference %.2e, %d strays" % (len(rows["user"]), n_diff, worst, st["stray_blocks"]))
    assert n_diff <= 1e-5 * len(rows["user"])


def test_rm2_netflix_24bit_matrix_against_fp32_matrix_all_rows(data, pruned):
    """the production matrix format against fp32 rows (FY_M24=0), all 48 M rows (see tests/test_full_size_gpu.py)"""
    rows32, _, st32 = run_rm2(data, TOPN, LAM, env={"FY_M24": "0"})
    assert st32["blocks_total"] == 0
    n_diff, worst = assert_same_lists(pruned[0], rows32, score_rtol=1e-5, tie_rtol=2e-5)
    print("24-bit matrix vs fp32 matrix: %d rows, %d differ (ties at a cut-off), worst relative score difference %.2e" % (len(rows32["user"]), n_diff, worst))
    assert n_diff <= 1e-4 * len(rows32["user"])


def test_itemsim_netflix_shape(data):
    check_itemsim(data, n_rows=4)


def test_itemsim_symmetric_build_equals_row_build_all_rows(data):
    n, n_diff, worst, st = compare_itemsim_builds(data)
    print("item similarity, symmetric build vs row-at-a-time build: %d rows, %d positions with another item (ties), worst similarity difference %.2e; "
          "%d candidates appended by the sweep, %d rows redone exactly" % (n, n_diff, worst, st["isim_candidates"], st["isim_redone_rows"]))
