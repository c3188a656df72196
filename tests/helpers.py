"""Shared by the CPU and GPU parity tests."""
import argparse


def settings_ns(d):
    return argparse.Namespace(**d)


def outcome(fn, seq, settings):
    """('ok', rows) or (exception name, None) -- the shape the golden fixtures store."""
    try:
        rows = fn(seq, settings_ns(settings))
        return "ok", [[s, e, m] for s, e, m in rows]
    except (AssertionError, IndexError, ValueError) as exc:
        return type(exc).__name__, None


def expected(case):
    return case.get("status", "ok"), case.get("rows")
