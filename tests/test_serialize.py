"""Demo on-disk formats: `imitation_amd.serialize` against the reference's own
`data/serialize.py` (run live under the oracle shim when /root/reference is present) and against
HuggingFace-directory fixtures written BY the reference (`tests/golden/hf_rollout*`)."""
import os
import warnings

import numpy as np
import pytest

from imitation_amd import data_types as dt
from imitation_amd import serialize as ser
from oracle import ref_shim

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def seeded_trajectories(with_infos: bool = False):
    """The trajectories behind `tests/golden/hf_rollout` (see make_golden.py)."""
    rng = np.random.default_rng(42)
    out = []
    for k, L in enumerate((5, 3, 8, 1)):
        infos = [{"step": i, "tag": "s%d" % k} for i in range(L)] if with_infos else None
        out.append(dt.TrajectoryWithRew(obs=rng.standard_normal((L + 1, 4)).astype(np.float32),
                                        acts=rng.integers(0, 2, L), rews=rng.standard_normal(L), infos=infos,
                                        terminal=bool(L % 2)))
    return out


def _same(a, b, infos=None):
    assert np.array_equal(np.asarray(a.obs), np.asarray(b.obs)) and np.array_equal(np.asarray(a.acts), np.asarray(b.acts))
    assert np.array_equal(np.asarray(a.rews), np.asarray(b.rews)) and bool(a.terminal) == bool(b.terminal)
    if infos is not None:
        assert list(b.infos) == infos


@pytest.mark.parametrize("name,with_infos", [("hf_rollout", False), ("hf_rollout_infos", True)])
def test_reads_directories_written_by_the_reference(name, with_infos):
    got = ser.load(os.path.join(GOLDEN, name))
    want = seeded_trajectories(with_infos)
    assert len(got) == len(want)
    for w, g in zip(want, got):
        _same(w, g, infos=(list(w.infos) if with_infos else [{}] * len(w.acts)))
        # the reference's decode goes through Python lists: float64 / int64 whatever was stored
        assert g.obs.dtype == np.float64 and g.acts.dtype == np.int64 and g.rews.dtype == np.float64
    assert [len(t) for t in got[1:3]] == [3, 8]   # slices work like the reference's sequence wrapper


def test_round_trip_and_transitions(tmp_path):
    trajs = seeded_trajectories(with_infos=True)
    ser.save(tmp_path / "d", trajs)
    back = ser.load(tmp_path / "d")
    for w, g in zip(trajs, back):
        _same(w, g, infos=list(w.infos))
    flat = ser.load_transitions(tmp_path / "d")
    assert len(flat) == 17 and flat.dones.sum() == sum(t.terminal for t in trajs)
    assert np.array_equal(flat.obs[:5], trajs[0].obs[:-1]) and np.array_equal(flat.next_obs[:5], trajs[0].obs[1:])
    assert len(ser.load_transitions(tmp_path / "d", n_max=6)) == 6


def test_non_json_infos_are_refused_not_mangled(tmp_path):
    bad = [dt.TrajectoryWithRew(obs=np.zeros((2, 1)), acts=np.zeros(1), rews=np.zeros(1),
                                infos=[{"terminal_observation": np.zeros(1)}], terminal=True)]
    with pytest.raises(TypeError, match="jsonpickle"):
        ser.save(tmp_path / "x", bad)
    with pytest.raises(ValueError, match="rewards but not all"):
        ser.trajectories_to_dict([seeded_trajectories()[0],
                                  type("T", (), dict(obs=np.zeros((2, 1)), acts=np.zeros(1), infos=None, terminal=True,
                                                     rews=None))()])


def test_legacy_npz_layout(tmp_path):
    """`serialize.py:50-67`: concatenated arrays + split points."""
    trajs = seeded_trajectories()
    lens = [len(t) for t in trajs]
    np.savez(tmp_path / "old.npz", obs=np.concatenate([t.obs for t in trajs]), acts=np.concatenate([t.acts for t in trajs]),
             rews=np.concatenate([t.rews for t in trajs]), infos=np.array([{}] * sum(lens)),
             terminal=np.array([t.terminal for t in trajs]), indices=np.cumsum(lens[:-1]))
    with pytest.warns(DeprecationWarning, match="old npz"):
        back = ser.load(tmp_path / "old.npz")
    assert len(back) == len(trajs)
    for w, g in zip(trajs, back):
        _same(w, g)
        assert g.obs.dtype == np.float32


@pytest.mark.reference
@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
def test_cross_read_with_live_reference(tmp_path):
    ref_shim.install()
    from imitation.data import serialize as rser
    from imitation.data import types as rtypes

    mine = seeded_trajectories(with_infos=True)
    theirs = [rtypes.TrajectoryWithRew(obs=t.obs, acts=t.acts, rews=t.rews, infos=np.array(list(t.infos)),
                                       terminal=t.terminal) for t in mine]
    rser.save(tmp_path / "ref", theirs)
    ser.save(tmp_path / "mine", mine)
    import datasets

    assert datasets.load_from_disk(str(tmp_path / "ref")).features == datasets.load_from_disk(str(tmp_path / "mine")).features
    for a, b, c in zip(rser.load(tmp_path / "mine"), ser.load(tmp_path / "ref"), rser.load(tmp_path / "ref")):
        _same(c, a, infos=list(c.infos))
        _same(c, b, infos=list(c.infos))
        assert a.obs.dtype == b.obs.dtype == c.obs.dtype and a.rews.dtype == b.rews.dtype == c.rews.dtype


@pytest.mark.reference
@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not present")
@pytest.mark.parametrize("fixture", ["cartpole_0", "pendulum_0"])
def test_reference_expert_fixtures_decode_identically(fixture):
    """The reference's own expert rollouts (`tests/testdata/expert_models/*/rollouts/final.npz`,
    legacy layout): same trajectories as its loader returns."""
    ref_shim.install()
    from imitation.data import serialize as rser

    path = f"/root/reference/tests/testdata/expert_models/{fixture}/rollouts/final.npz"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        theirs, mine = rser.load(path), ser.load(path)
    assert len(theirs) == len(mine) >= 56
    for a, b in zip(theirs, mine):
        _same(a, b)
        assert a.obs.dtype == b.obs.dtype and a.acts.dtype == b.acts.dtype
    flat = ser.load_transitions(path)
    assert len(flat) == sum(len(t.acts) for t in theirs)
