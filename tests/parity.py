"""Shared comparison rules for the parity tests.

Tolerance (stated by the task's north star): per-pixel hit keys must agree on
>= 99.99 % of pixels; radiance must match within 1e-4 relative OR 1 LSB of the
8-bit output, whose rule is clamp(int(v * 256), 0, 255).
"""
import numpy as np

KEY_AGREEMENT = 0.9999
REL_TOL = 1e-4


def to_8bit(rgba):
    with np.errstate(invalid="ignore"):
        v = np.nan_to_num(rgba * np.float32(256.0), nan=0.0, posinf=255.0, neginf=0.0)
    return np.clip(np.trunc(v), 0, 255).astype(np.int32)


def compare(gpu_rgba, gpu_keys, ref_rgba, ref_keys):
    """Returns a dict of agreement figures; NaN == NaN counts as equal."""
    out = {}
    if gpu_keys is not None:
        same = (gpu_keys == ref_keys).all(-1)
        out["key_agreement"] = float(same.mean())
        out["key_mismatches"] = int((~same).sum())
    both_nan = np.isnan(gpu_rgba) & np.isnan(ref_rgba)
    one_nan = np.isnan(gpu_rgba) ^ np.isnan(ref_rgba)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(gpu_rgba - ref_rgba) / np.maximum(np.abs(ref_rgba), np.float32(1e-12))
    rel[both_nan] = 0
    rel[gpu_rgba == ref_rgba] = 0
    lsb = np.abs(to_8bit(gpu_rgba) - to_8bit(ref_rgba))
    ok = (rel <= REL_TOL) | (lsb <= 1)
    ok &= ~one_nan
    out["max_rel"] = float(np.nanmax(rel))
    out["max_lsb"] = int(lsb.max())
    out["radiance_ok_fraction"] = float(ok.all(-1).mean())
    out["bit_exact_fraction"] = float(((gpu_rgba == ref_rgba) | both_nan).all(-1).mean())
    return out


def assert_parity(gpu_rgba, gpu_keys, ref_rgba, ref_keys, radiance_fraction=1.0):
    r = compare(gpu_rgba, gpu_keys, ref_rgba, ref_keys)
    if gpu_keys is not None:
        assert r["key_agreement"] >= KEY_AGREEMENT, r
    assert r["radiance_ok_fraction"] >= radiance_fraction, r
    return r
