"""Texel samplers (load time): the NumPy restatements in jrender_amd/io/obj.py are BIT-EXACT
against the reference's sampler kernels — through golden vectors generated from those kernels compiled for the
host (tests/golden/make_golden_textures.py), and, where /root/reference is mounted, directly on the spot cow's
texture (BASELINE configs C1/C2: data/obj/spot/spot_triangulated.obj, texture_res=5)."""
import os

import numpy as np
import pytest

from jrender_amd.io import obj as jio
from tests.util import bits_equal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "textures_golden.npz")
SPOT = "/root/reference/data/obj/spot/spot_triangulated.obj"


def test_softras_sampler_bit_exact_vs_golden():
    z = np.load(GOLD)
    for R in (1, 3, 5):
        got = jio.sample_textures(z["image"], z["tc_in"], z["softras_R%d_in" % R], z["is_update"])
        assert bits_equal(got, z["softras_R%d_out" % R]), R


def test_n3mr_sampler_bit_exact_vs_golden():
    z = np.load(GOLD)
    for ts in (2, 4):
        for w in range(4):
            for b in range(2):
                tc = z["tc_in"] if w == 3 else z["tc_wild"]
                got = jio.sample_textures_n3mr(z["image"], tc, z["n3mr_ts%d_in" % ts], z["is_update"], w, bool(b))
                assert bits_equal(got, z["n3mr_ts%d_w%d_b%d_out" % (ts, w, b)]), (ts, w, b)


def test_faces_without_update_keep_their_colour():
    z = np.load(GOLD)
    keep = z["is_update"] == 0
    assert keep.any()
    got = jio.sample_textures(z["image"], z["tc_in"], z["softras_R3_in"], z["is_update"])
    assert bits_equal(got[keep], z["softras_R3_in"][keep])
    got = jio.sample_textures_n3mr(z["image"], z["tc_in"], z["n3mr_ts2_in"], z["is_update"], 0, True)
    assert bits_equal(got[keep], z["n3mr_ts2_in"][keep])


@pytest.mark.skipif(not os.path.exists(SPOT), reason="reference data not mounted")
def test_spot_cow_textures_match_reference_kernels():
    from oracle import TexturesOracle
    o = TexturesOracle()
    v, f, tex, _, _, tc = jio.load_obj(SPOT, load_texture=True, texture_res=5)
    assert f.shape == (5856, 3) and tex.shape == (5856, 25, 3) and tc.shape == (5856, 3, 2)
    image = jio._material_image(SPOT, "spot_texture.png")
    upd = np.ones(f.shape[0], np.int32)
    ref = o.softras(image, tc, np.ones_like(tex), upd)
    assert bits_equal(tex, ref)                                      # the loader's result IS the sampler's output
    # the n3mr cube loader on the same mesh (REPEAT + bilinear, the defaults): faces with an exactly integer
    # texture coordinate are excluded (the reference oscillates on them, jrender_amd/io/obj.py)
    v2, f2, cube = jio.load_obj(SPOT, load_texture=True, texture_res=4, dr_type='n3mr')
    assert cube.shape == (5856, 4, 4, 4, 3) and bits_equal(f2, f)
    refc = o.n3mr(image, tc, np.full_like(cube, 0.5), upd, 0, True)
    ok = ~(tc == np.round(tc)).any((1, 2))
    assert ok.mean() > 0.99 and bits_equal(cube[ok], refc[ok])
    m = jio.__dict__  # noqa: F841
    import jrender_amd as jr
    mesh = jr.Mesh.from_obj(SPOT, load_texture=True, texture_res=4, dr_type='n3mr')
    assert mesh.textures.shape[-5:] == (5856, 4, 4, 4, 3)
