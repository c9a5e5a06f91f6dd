"""CPU: the two knn oracles (float32 brute force, float64 k-d tree) agree; hand-computed known answers."""
import numpy as np

from oracle import knn_oracle


def test_known_answer_unit_lattice():
    p = np.stack(np.meshgrid(*[np.arange(4, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    d = knn_oracle.dist2_bruteforce(p)
    assert np.all(d == 1.0)                      # every lattice point has >= 3 neighbours at distance 1


def test_known_answer_four_points():
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32)
    want = np.array([(1 + 4 + 9) / 3, (1 + 5 + 10) / 3, (4 + 5 + 13) / 3, (9 + 10 + 13) / 3], np.float32)
    np.testing.assert_allclose(knn_oracle.dist2_bruteforce(p), want, rtol=1e-6)
    np.testing.assert_allclose(knn_oracle.dist2_kdtree(p), want, rtol=1e-6)


def test_bruteforce_matches_kdtree_with_duplicates():
    rng = np.random.default_rng(0)
    p = rng.normal(size=(3000, 3)).astype(np.float32)
    p[10] = p[11] = p[12]
    a, b = knn_oracle.dist2_bruteforce(p), knn_oracle.dist2_kdtree(p)
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-12)


def test_fewer_than_four_points():
    p = np.array([[0, 0, 0], [3, 4, 0]], np.float32)
    np.testing.assert_allclose(knn_oracle.dist2_bruteforce(p), [25, 25])
    assert knn_oracle.dist2_bruteforce(p[:1])[0] == 0
