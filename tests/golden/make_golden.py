#!/usr/bin/env python3
"""Transcribe the reference's own golden test vectors for the RM2 path into JSON.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

Source of every array (data values only, no code is copied):
  T = /root/reference/src/test/java/es/udc/fi/dc/irlab/testdata/
  * T/RMTestData.java:25-27    numberOfUsers / numberOfItems / numberOfClusters
  * T/RMTestData.java:32-232   A  (100 items x 30 users, A[i][j] = rating of user j+1 for item i+1)
  * T/RMTestData.java:234-403  recommendations  (507 (user, item, score) triples)
  * T/RMTestData.java:408-410  userSum,  :415-421 itemSum,  :426 totalSum,  :431-464 itemColl
  * T/ClusteringTestData.java:90-93  clustering (cluster of user u at [u-1]), clusteringCount
  * T/RMTestData2.java:25-60   the 3x5 toy (A, userSum, itemSum, totalSum, itemColl, clustering)
  * T/ClusteringTestData.java:28-93   H (30 users x 5 clusters) -> clustering, clusteringCount   (cluster assignment stage,
    asserted by T/../nmf/clustering/TestClusterAssignment.java:43-69)
  * T/NMFTestData.java / T/PPCTestData.java   A (100 items x 30 users), W_init, H_init -> W_one, H_one (1 iteration), W_ten, H_ten
    (10 iterations); PPC's 5 x 7 toy Ap, h0p, w0p -> w1p, h1p
  * T/SubClusteringTestData.java:25-100  H0 (users 1..17), H1 (users 18..30), numberOfSubClusters -> clustering
    (sub-cluster assignment, TestClusterAssignment.java:71-103)
Parameters of the reference integration test that produced `recommendations`
(T/../rm/TestHDFSRM2.java:39-75 with T/../util/HadoopIntegrationTest.java:81-100):
  lambda=0.5, clusterSplit=5, splitSize=3, numberOfRecommendations=1000, tolerance 1e-4 absolute.
The reference is Apache-2.0 (/root/reference/LICENSE.txt); the fixtures keep that attribution.
"""
import json
import os
import re
import sys

T = "/root/reference/src/test/java/es/udc/fi/dc/irlab/testdata/"
HERE = os.path.dirname(os.path.abspath(__file__))


def java_array(text, name):
    """Return the Java array literal assigned to `name` as nested Python lists."""
    m = re.search(r"\b%s\s*=\s*new\s+\w+\s*(?:\[\s*\])+\s*(\{.*?\})\s*;" % re.escape(name), text, re.S)
    if not m:
        raise KeyError(name)
    lit = m.group(1)
    lit = re.sub(r"(\d)[dD](?=[\s,}])", r"\1", lit)    # "2.585d" -> "2.585"
    lit = re.sub(r"(\d)\.(?=[\s,}])", r"\1.0", lit)  # "238." -> "238.0"
    lit = lit.replace("{", "[").replace("}", "]")
    return json.loads(lit)


def java_scalar(text, name):
    m = re.search(r"\b%s\s*=\s*([-+0-9.eE]+)\s*;" % re.escape(name), text)
    return float(m.group(1))


def main():
    if not os.path.isdir(T):
        sys.exit("reference tree not present; fixtures are already committed")
    rm = open(T + "RMTestData.java").read()
    cl = open(T + "ClusteringTestData.java").read()
    rm2 = open(T + "RMTestData2.java").read()
    attribution = ("values transcribed from dvalcarce/filmyou-core (Apache-2.0) "
                   "src/test/java/es/udc/fi/dc/irlab/testdata/")

    out = {
        "_source": attribution + "RMTestData.java + ClusteringTestData.java",
        "params": {"lambda": 0.5, "clusterSplit": 5, "splitSize": 3,
                   "numberOfRecommendations": 1000, "filterUsers": 0,
                   "reference_tolerance_abs": 1e-4},
        "numberOfUsers": int(java_scalar(rm, "numberOfUsers")),
        "numberOfItems": int(java_scalar(rm, "numberOfItems")),
        "numberOfClusters": int(java_scalar(rm, "numberOfClusters")),
        "A_items_by_users": java_array(rm, "A"),
        "recommendations": java_array(rm, "recommendations"),
        "userSum": java_array(rm, "userSum"),
        "itemSum": java_array(rm, "itemSum"),
        "totalSum": java_scalar(rm, "totalSum"),
        "itemColl": java_array(rm, "itemColl"),
        "clustering": java_array(cl, "clustering"),
        "clusteringCount": java_array(cl, "clusteringCount"),
    }
    assert len(out["A_items_by_users"]) == 100 and len(out["A_items_by_users"][0]) == 30
    assert len(out["recommendations"]) == 507
    assert len(out["clustering"]) == 30 and sum(out["clusteringCount"]) == 30
    with open(os.path.join(HERE, "rm_test_data.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))

    sub = open(T + "SubClusteringTestData.java").read()

    def java_array2(text, name):   # `H1 = { ... }` is written without `new double[][]`
        m = re.search(r"\b%s\s*=\s*(?:new\s+\w+\s*(?:\[\s*\])+\s*)?(\{.*?\})\s*;" % re.escape(name), text, re.S)
        lit = re.sub(r"(\d)\.(?=[\s,}])", r"\1.0", m.group(1)).replace("{", "[").replace("}", "]")
        return json.loads(lit)

    clus = {
        "_source": attribution + "ClusteringTestData.java + SubClusteringTestData.java",
        "numberOfUsers": int(java_scalar(cl, "numberOfUsers")),
        "numberOfClusters": int(java_scalar(cl, "numberOfClusters")),
        "H": java_array(cl, "H"),
        "clustering": java_array(cl, "clustering"),
        "clusteringCount": java_array(cl, "clusteringCount"),
        "sub": {"numberOfUsers": int(java_scalar(sub, "numberOfUsers")), "numberOfClusters": int(java_scalar(sub, "numberOfClusters")),
                "numberOfSubClusters": int(java_scalar(sub, "numberOfSubClusters")),
                "H0": java_array2(sub, "H0"), "H0_first_user": 1, "H1": java_array2(sub, "H1"), "H1_first_user": 18,
                "clustering": java_array2(sub, "clustering")},
    }
    assert len(clus["H"]) == 30 and len(clus["H"][0]) == 5
    assert len(clus["sub"]["H0"]) + len(clus["sub"]["H1"]) == 30
    with open(os.path.join(HERE, "clustering_test_data.json"), "w") as f:
        json.dump(clus, f, separators=(",", ":"))

    nmf, ppc = open(T + "NMFTestData.java").read(), open(T + "PPCTestData.java").read()
    fact = {
        "_source": attribution + "NMFTestData.java + PPCTestData.java (asserted by T/../nmf/NMFHDFSDriverTest.java:36-70, "
                   "T/../nmf/ppc/PPCHDFSDriverTest.java, T/../nmf/hcomputation/*, T/../nmf/wcomputation/* with accuracy 1e-4)",
        "nmf": {k: java_array(nmf, k) for k in ("A", "W_init", "H_init", "W_one", "H_one", "W_ten", "H_ten")},
        "ppc": {k: java_array(ppc, k) for k in ("Ap", "h0p", "w0p", "w1p", "h1p", "A", "W_init", "H_init", "W_one", "H_one", "W_ten", "H_ten")},
    }
    assert len(fact["nmf"]["A"]) == 100 and len(fact["nmf"]["H_ten"]) == 30 and len(fact["nmf"]["W_ten"][0]) == 10
    with open(os.path.join(HERE, "factorization_test_data.json"), "w") as f:
        json.dump(fact, f, separators=(",", ":"))

    toy = {
        "_source": attribution + "RMTestData2.java",
        "numberOfUsers": int(java_scalar(rm2, "numberOfUsers")),
        "numberOfItems": int(java_scalar(rm2, "numberOfItems")),
        "numberOfClusters": int(java_scalar(rm2, "numberOfClusters")),
        "A_items_by_users": java_array(rm2, "A"),
        "userSum": java_array(rm2, "userSum"),
        "itemSum": java_array(rm2, "itemSum"),
        "totalSum": java_scalar(rm2, "totalSum"),
        "itemColl": java_array(rm2, "itemColl"),
        "clustering": java_array(rm2, "clustering"),
        "clusteringCount": java_array(rm2, "clusteringCount"),
    }
    with open(os.path.join(HERE, "rm_test_data2.json"), "w") as f:
        json.dump(toy, f, separators=(",", ":"))
    print("wrote rm_test_data.json (%d recs) and rm_test_data2.json" % len(out["recommendations"]))


if __name__ == "__main__":
    main()
