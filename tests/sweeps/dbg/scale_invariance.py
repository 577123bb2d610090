"""Is R(4A) = 4 R(A) bit for bit?  Dense fronts of a few shapes, then a small grid problem front by front."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
from tests.test_dense_gpu import run_dense
rng = np.random.default_rng(1)
for shape in [(40, 20), (300, 100), (256, 32), (257, 32), (600, 33), (1000, 400)]:
    A = rng.uniform(-1, 1, size=shape)
    (R1,), _ = run_dense([A])
    (R4,), _ = run_dense([A * 4.0])
    k = min(shape)
    d = np.triu(R4[:k]) - 4.0 * np.triu(R1[:k])
    print(shape, "entries that differ:", int(np.count_nonzero(d)), "max rel %.2e" % (np.abs(d).max() / np.abs(R1).max()))
    if np.count_nonzero(d):
        r, c = np.nonzero(d)
        print("   first differing row", r.min(), "col", c[r == r.min()].min())
# integers and sqrt(0.1), as the grid problems have
A = rng.integers(-1, 7, size=(300, 64)).astype(float)
A[200:264, :] += np.sqrt(0.1) * np.eye(64)
(R1,), _ = run_dense([A]); (R4,), _ = run_dense([A * 4.0])
d = np.triu(R4[:64]) - 4.0 * np.triu(R1[:64])
print("integer-like", "entries that differ:", int(np.count_nonzero(d)))
from tests.util import stair_front
for shape, ag in [((202, 153), 4), ((400, 150), 4), ((1200, 300), 8)]:
    A = stair_front(shape[0], shape[1], rng, aggression=ag)
    (R1,), _ = run_dense([A]); (R4,), _ = run_dense([A * 4.0])
    k = min(shape)
    d = np.triu(R4[:k]) - 4.0 * np.triu(R1[:k])
    print("staircase", shape, "entries that differ:", int(np.count_nonzero(d)), "max rel %.2e" % (np.abs(d).max() / np.abs(R1).max()))
# values spread over many magnitudes, as fronts deep in a tree have
A = rng.uniform(-1, 1, size=(202, 153)) * 10.0 ** rng.integers(-8, 1, size=(202, 1))
(R1,), _ = run_dense([A]); (R4,), _ = run_dense([A * 4.0])
d = np.triu(R4[:153]) - 4.0 * np.triu(R1[:153])
print("graded rows", "entries that differ:", int(np.count_nonzero(d)), "max rel %.2e" % (np.abs(d).max() / np.abs(R1).max()))
