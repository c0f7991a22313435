"""Shared helpers for the parity tests."""
import numpy as np


def to_np(t):
    return t.detach().float().cpu().numpy()


def report(name, got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    diff = np.abs(got - ref)
    i = int(np.argmax(diff)) if diff.size else 0
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    return (f"{name}: shape {got.shape} max|diff|={diff.max() if diff.size else 0:.3e} at flat {i} "
            f"(got {got.flat[i] if diff.size else 0:.6g} ref {ref.flat[i] if diff.size else 0:.6g}) "
            f"ref max|.|={scale:.3e} mean|diff|={diff.mean() if diff.size else 0:.3e} "
            f"nan(got)={int(np.isnan(got).sum())}")


def assert_close(name, got, ref, atol, rtol=0.0):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    msg = report(name, got, ref)
    print(msg)
    assert np.all(np.isfinite(got)), msg
    assert np.all(np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= atol + rtol * np.abs(ref)), msg
