"""CRF restatement pinned by brute-force enumeration (pytorch-crf itself is absent: parity unpinned
against the package, pinned against the mathematics)."""
import numpy as np
import pytest
import torch

from oracle import mtvaf_oracle as O


def _rand_crf(seed, B, S, C, lengths):
    g = torch.Generator().manual_seed(seed)
    em = torch.randn(B, S, C, generator=g)
    start, end = torch.rand(C, generator=g) - 0.5, torch.rand(C, generator=g) - 0.5
    trans = torch.rand(C, C, generator=g) - 0.5
    mask = torch.zeros(B, S, dtype=torch.uint8)
    for b, L in enumerate(lengths):
        mask[b, :L] = 1
    tags = torch.randint(0, C, (B, S), generator=g)
    return em, tags, mask, start, end, trans


@pytest.mark.parametrize("C,lengths", [(11, [4, 3, 1, 2]), (5, [6, 6, 2]), (3, [7, 1])])
def test_crf_matches_bruteforce(C, lengths):
    B, S = len(lengths), max(lengths) + 1  # one always-padded column
    em, tags, mask, start, end, trans = _rand_crf(7 + C, B, S, C, lengths)
    logZ, best = O.crf_bruteforce(em, mask, start, end, trans)
    got = O.crf_log_partition(em, mask, start, end, trans)
    np.testing.assert_allclose(got.numpy(), np.array(logZ), rtol=1e-5, atol=1e-5)
    assert O.crf_decode(em, mask, start, end, trans) == best
    # gold score: explicit sum
    sc = O.crf_sequence_score(em, tags, mask, start, end, trans)
    for b, L in enumerate(lengths):
        s = start[tags[b, 0]] + em[b, 0, tags[b, 0]]
        for t in range(1, L):
            s = s + trans[tags[b, t - 1], tags[b, t]] + em[b, t, tags[b, t]]
        s = s + end[tags[b, L - 1]]
        assert abs(float(s) - float(sc[b])) < 1e-5
    llh = O.crf_log_likelihood(em, tags, mask, start, end, trans, "mean")
    assert abs(float(llh) - float((sc - got).mean())) < 1e-6
    assert float(llh) < 0


def test_crf_probabilities_sum_to_one():
    em, tags, mask, start, end, trans = _rand_crf(3, 1, 3, 4, [3])
    import itertools
    tot = 0.0
    for path in itertools.product(range(4), repeat=3):
        t = torch.tensor([list(path)])
        tot += float(torch.exp(O.crf_log_likelihood(em, t, mask, start, end, trans, "none"))[0])
    assert abs(tot - 1.0) < 1e-5
