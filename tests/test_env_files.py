"""Environment file reader / writer (msgpack layout of the reference's test-envs)."""
import glob
import os

import numpy as np
import pytest

from tests.golden_util import env_names, load_env

REF = "/root/reference/test-envs"


def test_round_trip(tmp_path):
    from flatland_marl_amd.envs import load_env_file, save_env_file
    for name in ("Test_0_Level_0", "Test_6_Level_1", "Gen_30x45_2"):
        fx = load_env(name)
        p = os.path.join(tmp_path, name + ".pkl")
        save_env_file(p, fx)
        back = load_env_file(p)
        for k in ("grid", "initial_position", "initial_direction", "direction", "target", "speed", "handle", "status",
                  "malfunction"):
            assert np.array_equal(back[k], fx[k]), (name, k)
            assert back[k].dtype == fx[k].dtype, (name, k)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's files exist in the build container only")
def test_reads_the_reference_files():
    from flatland_marl_amd.envs import load_env_file
    files = sorted(glob.glob(os.path.join(REF, "Test_*", "Level_*.pkl")))
    assert len(files) == len([n for n in env_names() if n.startswith("Test_")])
    for f in files:
        name = "%s_%s" % (os.path.basename(os.path.dirname(f)), os.path.basename(f)[:-4])
        got, want = load_env_file(f), load_env(name)
        for k in want:
            assert np.array_equal(got[k], want[k]), (name, k)
