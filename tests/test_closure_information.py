"""Loop-closure covariance / information (fast_correlative_scan_matcher_2d.cc:537-561,
submap_loop_closer.cpp:276): known-answer tests of the oracle on CPU, parity of
sgo_closure_information with the oracle on the GPU."""
import numpy as np
import pytest

from oracle import np_oracle as no
from sparse_gslam_amd import capi


def _window(x0=3, y0=-7, k0=20, sw=5, w=5, nap=20, res=0.05, step=0.0035):
    return dict(x_index_offset=x0, y_index_offset=y0, scan_index=k0, scan_window=sw, w_size=w,
                num_angular_perturbations=nap, resolution=res, angular_step=step)


def _n_scores(win):
    return (2 * win["w_size"] + 1) ** 2 * (2 * win["scan_window"] + 1)


def test_uniform_scores_give_the_moments_of_the_grid():
    # sum over m = -w..w of m^2 / (2w+1) = w (w+1) / 3: with equal scores the covariance is diagonal
    win = _window()
    cov, info = no.closure_information(win, np.full(_n_scores(win), 0.5, np.float32))
    w, sw, res, step = win["w_size"], win["scan_window"], win["resolution"], win["angular_step"]
    want = np.diag([res**2 * w * (w + 1) / 3, res**2 * w * (w + 1) / 3, step**2 * sw * (sw + 1) / 3])
    assert np.abs(cov - want).max() < 1e-15
    assert np.abs(info @ cov - np.eye(3)).max() < 1e-9


def test_offsets_do_not_change_the_covariance_and_axes_are_swapped_as_in_the_reference():
    # x of the pose comes from the y cell offset and vice versa (correlative_scan_matcher_2d.h:78-82)
    rng = np.random.default_rng(0)
    win0 = _window(x0=0, y0=0, k0=20)
    sc = rng.uniform(0.1, 1.0, _n_scores(win0)).astype(np.float32)
    cov0, _ = no.closure_information(win0, sc)
    cov1, _ = no.closure_information(_window(x0=40, y0=-25, k0=9, nap=30), sc)
    assert np.abs(cov0 - cov1).max() < 1e-12
    # a score ridge along the i (x cell) axis must show up as variance of the pose's y component
    w = win0["w_size"]
    ridge = np.zeros((2 * w + 1, 2 * w + 1, 2 * win0["scan_window"] + 1), np.float32)
    ridge[:, w, win0["scan_window"]] = 1.0
    cov, info = no.closure_information(win0, ridge.ravel())
    assert cov[1, 1] > 1e-4 and abs(cov[0, 0]) < 1e-18 and abs(cov[2, 2]) < 1e-18
    assert not np.isfinite(info).all()            # singular, as covariance.inverse() would give


def test_zero_scores_are_not_finite():
    win = _window(sw=0)
    cov, info = no.closure_information(win, np.zeros(_n_scores(win), np.float32))
    assert not np.isfinite(cov).any() and not np.isfinite(info).any()


def _random_batch(rng, n):
    wins, scores = [], []
    for q in range(n):
        k0 = int(rng.integers(0, 41))
        sw = min(5, k0, 40 - k0)                  # the reference's scan_window rule (:541)
        win = _window(x0=int(rng.integers(-60, 60)), y0=int(rng.integers(-60, 60)), k0=k0, sw=sw,
                      w=5 if q % 7 else int(rng.integers(0, 9)), nap=20,
                      res=float(rng.choice([0.05, 0.1])), step=float(rng.uniform(0.002, 0.01)))
        # a score bump around a random cell plus a floor, like a match response
        w = win["w_size"]
        ii, jj, kk = np.meshgrid(np.arange(-w, w + 1), np.arange(-w, w + 1), np.arange(-sw, sw + 1), indexing="ij")
        c = rng.uniform(-2, 2, 3)
        s = 0.15 + 0.8 * np.exp(-((ii - c[0]) ** 2 + (jj - c[1]) ** 2) / rng.uniform(2, 12) - (kk - c[2]) ** 2 / 6.0)
        wins.append(win)
        scores.append(s.astype(np.float32).ravel())
    return wins, scores


@pytest.mark.gpu
def test_gpu_matches_oracle_on_ragged_batch():
    rng = np.random.default_rng(11)
    wins, scores = _random_batch(rng, 301)
    with capi.Optimizer(0) as opt:
        cov, info = opt.closure_information(wins, np.concatenate(scores))
    for q, (win, sc) in enumerate(zip(wins, scores)):
        oc, oi = no.closure_information(win, sc)
        # fp64 sums in a different order, then a difference of nearly equal numbers (offsets of up
        # to 60 cells against a spread of a few cells): 1e-9 of the largest second moment
        scale = max(np.abs(oc).max(), (60 * win["resolution"]) ** 2 * 1e-3)
        assert np.abs(cov[q] - oc).max() <= 1e-9 * scale, q
        if win["w_size"] > 0 and win["scan_window"] > 0:
            assert np.isfinite(oi).all()
            assert np.abs(info[q] - oi).max() <= 1e-6 * np.abs(oi).max(), q
            assert np.abs(info[q] @ cov[q] - np.eye(3)).max() < 1e-6
        assert np.array_equal(cov[q], cov[q].T) and np.array_equal(info[q], info[q].T, equal_nan=True)


@pytest.mark.gpu
def test_gpu_degenerate_windows_and_argument_checks():
    win = _window(sw=0)
    with capi.Optimizer(0) as opt:
        cov, info = opt.closure_information([win], np.zeros(_n_scores(win), np.float32))
        assert not np.isfinite(cov).any() and not np.isfinite(info).any()
        cov, info = opt.closure_information([], np.zeros(0, np.float32))
        assert cov.shape == (0, 3, 3)
        with pytest.raises(capi.SgoError, match="reaches past"):
            opt.closure_information([_window()], np.ones(10, np.float32))
        with pytest.raises(capi.SgoError, match="negative or oversized"):
            opt.closure_information([dict(_window(), score_offset=-5)], np.ones(2000, np.float32))
        # results feed straight into a closure edge: information of a sane window is SPD
        wins, scores = _random_batch(np.random.default_rng(3), 8)
        _, info = opt.closure_information(wins[1:2], scores[1])
        assert np.all(np.linalg.eigvalsh(info[0]) > 0)
