"""Oracle observation restatements vs golden vectors from the reference featurizers
(src/features/component.py, src/features/model_ready.py)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden

FEATS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "feat_*.npz")))


def oracle_at_states(oracle_mod, g):
    meta = g["meta"]
    S = len(g["pos"])
    ob = oracle_mod.OracleBatch(oracle_mod.config_from_fixture_meta(meta), S)
    for s in range(S):
        kw = dict(pos=g["pos"][s], alive=g["alive"][s], imp_mask=g["imp"][s])
        if g["jobpos"].shape[1]:
            kw.update(jobpos=g["jobpos"][s], jobdone=g["jobdone"][s])
        ob.set_state(s, **kw)
    return ob


@pytest.mark.parametrize("name", FEATS)
def test_oracle_observations_match_reference(oracle_mod, name):
    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    ob = oracle_at_states(oracle_mod, g)
    np.testing.assert_array_equal(ob.obs_raw(), g["raw"])
    for comp in g["meta"]["flat"]:
        got = ob.obs_flat([comp])
        want = g["flat_" + comp]
        assert got.dtype == np.float32 and got.shape == want.shape, comp
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), comp
    # a composite is the concatenation in list order (component.py:146-149)
    comps = g["meta"]["flat"]
    np.testing.assert_array_equal(ob.obs_flat(comps), np.concatenate([g["flat_" + c] for c in comps], axis=1))
    if "planes_spatial" in g:
        sp, non = ob.obs_planes()
        np.testing.assert_array_equal(sp, g["planes_spatial"])
        # GlobalFeaturizer non-spatial = [alive, job_status] (model_ready.py:237-247)
        np.testing.assert_array_equal(non, g["planes_non_spatial"])
        # PerspectiveFeaturizer = channel rotation of the same planes (model_ready.py:184-215)
        A = ob.A
        for i in range(A):
            order = list(range(sp.shape[1]))
            agents = list(range(A))
            order[0] = agents[0] = i
            for k in range(1, i + 1):
                order[k] = agents[k] = k - 1
            np.testing.assert_array_equal(sp[:, order], g["persp_spatial"][:, i])
            want_ns = np.concatenate([non[:, :A][:, agents], non[:, A:]], axis=1)
            np.testing.assert_array_equal(want_ns, g["persp_non_spatial"][:, i])
