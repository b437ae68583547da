"""The oracle's material arithmetic against a THIRD statement of it (tests/disney_f64.py: float64 numpy written from the
published formulas, not from the oracle's or the kernel's lines). The GPU-vs-oracle tests cannot see a term that was
mis-transcribed on both sides in the same way (28 % of shading.h is textually the oracle's); this one can.

10 000 random direction pairs per material, all lobes and their mixes, both transport modes; relative tolerance 1e-5 of
the value's own scale for f and the pdfs (99 % of the pairs; 1e-4 for the rest) away from grazing directions
(|cos| < 0.02, where the float32 oracle and a float64 formula legitimately part), 2e-4 everywhere."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from stratum_amd import wire

from disney_f64 import connection_dvc, disney_eval, power_heuristic, shading_normal_correction

MATERIALS = {
    "diffuse": dict(base_color=(0.73, 0.45, 0.2), emission=0.0, metallic=0.0, roughness=0.0, anisotropic=0.0, subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0, transmission=0.0, eta=1.5),
    "rough diffuse + subsurface": dict(base_color=(0.5, 0.6, 0.7), emission=0.0, metallic=0.0, roughness=0.7, anisotropic=0.0, subsurface=0.6, clearcoat=0.0, clearcoat_gloss=0.0, transmission=0.0, eta=1.5),
    "metal": dict(base_color=(0.9, 0.7, 0.4), emission=0.0, metallic=1.0, roughness=0.35, anisotropic=0.0, subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0, transmission=0.0, eta=1.5),
    "anisotropic metal": dict(base_color=(0.8, 0.8, 0.9), emission=0.0, metallic=1.0, roughness=0.5, anisotropic=0.8, subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0, transmission=0.0, eta=1.5),
    "glass": dict(base_color=(0.95, 0.97, 0.9), emission=0.0, metallic=0.0, roughness=0.3, anisotropic=0.0, subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0, transmission=1.0, eta=1.45),
    "clearcoated plastic": dict(base_color=(0.2, 0.5, 0.3), emission=0.0, metallic=0.0, roughness=0.5, anisotropic=0.0, subsurface=0.1, clearcoat=1.0, clearcoat_gloss=0.7, transmission=0.0, eta=1.5),
    "everything at once": dict(base_color=(0.6, 0.4, 0.5), emission=0.0, metallic=0.35, roughness=0.45, anisotropic=0.4, subsurface=0.3, clearcoat=0.6, clearcoat_gloss=0.2, transmission=0.5, eta=1.33),
    "emitter": dict(base_color=(1.0, 0.7, 0.2), emission=12.0, metallic=0.0, roughness=0.0, anisotropic=0.0, subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0, transmission=0.0, eta=0.0),
}


def record(p):
    rec = np.zeros(1, wire.MaterialRecord)
    rec["values"]["value"][0, 0] = (*p["base_color"], p["emission"])
    rec["values"]["value"][0, 1] = (p["metallic"], p["roughness"], p["anisotropic"], p["subsurface"])
    rec["values"]["value"][0, 2] = (p["clearcoat"], p["clearcoat_gloss"], p["transmission"], p["eta"])
    rec["values"]["image_index"][0] = 0xFFFFFFFF
    rec["alpha_mask_index"] = 0xFFFFFFFF
    rec["bump_index"] = 0xFFFFFFFF
    return rec


def directions(n, seed):
    rng = np.random.RandomState(seed)
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("name", list(MATERIALS))
@pytest.mark.parametrize("adjoint", [False, True])
def test_disney_eval_against_the_float64_formulas(name, adjoint):
    p = MATERIALS[name]
    wi, wo = directions(10000, 1), directions(10000, 2)
    got = orc.disney_eval_adjoint(record(p), wi, wo, adjoint).astype(np.float64)
    f, pf, pr = disney_eval(p, wi.astype(np.float64), wo.astype(np.float64), adjoint)
    ref = np.concatenate([f, pf[:, None], pr[:, None]], 1)
    assert np.isfinite(ref).all()
    easy = (np.abs(wi[:, 2]) > 0.02) & (np.abs(wo[:, 2]) > 0.02)
    scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max(axis=0, keepdims=True) + 1e-30)
    err = np.abs(got - ref) / scale
    print("%s adjoint=%s: max rel err %.2e (easy directions), %.2e (all)" % (name, adjoint, err[easy].max(), err.max()))
    # 1e-5 for all but the pairs that sit on a cancellation the formula itself has in binary32 — refraction next to the
    # critical angle (1 - sin^2 / eta^2), the clearcoat's GTR1 at its peak (1 + (a^2 - 1) h_z^2 with |h_z| -> 1): the
    # reference's own float32 shader loses the same digits there; those pairs stay within 1e-4
    assert np.quantile(err[easy], 0.99) < 1e-5
    assert err[easy].max() < 1e-4
    assert err.max() < 2e-4
    if p["emission"] > 0:
        assert not got.any()  # an emitter does not scatter (disney_material.hlsli:83,142-146)


def test_integrator_helpers_against_the_formulas():
    rng = np.random.RandomState(5)
    # power heuristic, path.hlsli:8-15, and the dVC recurrence, :31-38
    ab = rng.uniform(1e-3, 50, (10000, 2)).astype(np.float32)
    assert np.allclose(orc.mis(ab), power_heuristic(ab[:, 0].astype(np.float64), ab[:, 1].astype(np.float64)), rtol=1e-5, atol=0)
    abc = rng.uniform(1e-2, 20, (10000, 3)).astype(np.float32)
    for specular in (False, True):
        ref = connection_dvc(*(abc[:, k].astype(np.float64) for k in range(3)), specular)
        assert np.allclose(orc.connection_dvc(abc, specular), ref, rtol=1e-5, atol=0)
    # shading-normal correction, path.hlsli:67-98: the light-leak test, the shadow-terminator term, the adjoint ratio
    rows = rng.uniform(-1, 1, (10000, 5)).astype(np.float32)
    rows[np.abs(rows) < 0.05] = 0.3  # away from the divisions by ~0 where float32 and float64 legitimately part
    for fix in (False, True):
        for adjoint in (False, True):
            ref = shading_normal_correction(*(rows[:, k] for k in range(5)), shadow_fix=fix, adjoint=adjoint)
            got = orc.shading_normal_correction(rows, fix, adjoint)
            assert np.allclose(got, ref, rtol=2e-5, atol=1e-7), (fix, adjoint, np.abs(got - ref).max())
    assert (shading_normal_correction(*(rows[:, k] for k in range(5))) == 0).mean() > 0.3  # the leak test fires
