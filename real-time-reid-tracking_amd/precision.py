"""Arithmetic of the plugin surface (Extractor, build_model backbones): which of the library's three modes a model runs in.

    precision=None   -> $REID_PRECISION if set, else "f16x3" - the mode bench.py reports as `value`
    "f16x3" / 2      fp32-class: fp32 storage, every convolution / Linear as three f16 matrix-core products per multiply on
                     hi/lo-split operands, fp32 accumulate.  Held to the exact-fp32 mode's parity bar against the reference's
                     vectors (tests/test_gpu_parity.py, the [2] parametrisations).
    "f32" / 0        exact fp32 MFMA - the reference's arithmetic, and the C library's own default (include/reid_hip.h)
    "f16" / 1        fp16 storage / fp32 accumulate (north_star's 1e-3 cosine tolerance)

The fp32-class mode has an operand range (|w| 2^11 and every activation inside f16): a checkpoint whose weights are outside it is
refused by reid_ctx_set_precision(2) with REID_ERR_ARG, an activation outside it raises the context's sticky fault word
(REID_ERR_STATE).  The plugin objects then fall back to exact fp32 - once, with ONE log line - instead of failing the tracker:
`fallback_to_exact` below.  The reference has no such mode to mirror (feature_extractor.py:15-29 runs torch fp32 on cuda).
"""
import logging
import os

from ._ffi import ReidHipError

EXACT, F16, F32_CLASS = 0, 1, 2
_NAMES = {"f32": 0, "fp32": 0, "exact": 0, "0": 0, "f16": 1, "fp16": 1, "half": 1, "1": 1, "f16x3": 2, "fp32-class": 2, "2": 2}
LABEL = {0: "f32", 1: "f16", 2: "f16x3"}
DEFAULT = "f16x3"

log = logging.getLogger("root.tracker")


def resolve(precision=None):
    """precision argument (None, name or 0/1/2) -> mode 0/1/2; None reads $REID_PRECISION, then DEFAULT."""
    if precision is None:
        precision = os.environ.get("REID_PRECISION") or DEFAULT
    key = str(precision).strip().lower()
    if key not in _NAMES:
        raise ValueError("precision must be one of f32 / f16 / f16x3 (or 0 / 1 / 2), got %r" % (precision,))
    return _NAMES[key]


def refused(err):
    """True for the two ways the fp32-class mode turns a checkpoint down: weights it cannot split (REID_ERR_ARG from
    reid_ctx_set_precision) or an activation outside f16's range at run time (REID_ERR_STATE, bit 0 of the sticky fault word -
    NOT the "non-finite embedding" fault, which says nothing about the arithmetic)."""
    return isinstance(err, ReidHipError) and err.status in (-1, -3) and "fp32-class" in str(err)


def fallback_to_exact(owner, err, what):
    """The one log line of a fall back; ``owner._mode`` becomes exact fp32 for the rest of the object's life."""
    owner._mode = EXACT
    log.warning("%s: the fp32-class arithmetic (precision f16x3) cannot run this checkpoint, using exact fp32 from now on (%s)", what, err)


def run(owner, eng, what, fn):
    """``fn(eng)`` with the owner's weights bound and the context in ``owner._mode``; the (process-wide, shared) engine gets its
    previous mode back afterwards.  ``owner._needs_bind(eng)`` / ``owner._do_bind(eng)``: the weights are (re)loaded in mode 0,
    which accepts every checkpoint, then the mode is switched.  When the fp32-class mode turns THIS owner's checkpoint down (at the
    switch, or through the fault word while running) the owner falls back to exact fp32 for good and the call is repeated once.
    The engine is shared by both backbones: a refusal at the switch that is about the OTHER architecture's checkpoint
    (``eng.precision_ok(owner arch, 2)`` says this one is fine) costs this call its mode - it runs in exact fp32, logged once per
    owner - but not the owner's; and a fault word that was already up when the call began belongs to somebody else's work and is
    re-raised, not cleared."""
    arch = 1 if getattr(owner, "_arch", "seres18") == "swin" else 0
    for _ in range(2):
        prev = eng.precision
        fault_before = eng.fault_bits() if hasattr(eng, "fault_bits") else 0
        mode = owner._mode
        try:
            if owner._needs_bind(eng):
                if prev != EXACT:
                    eng.set_precision(EXACT)
                owner._do_bind(eng)
            if eng.precision != mode:
                try:
                    eng.set_precision(mode)
                except ReidHipError as e:
                    if mode == F32_CLASS and e.status == -1 and hasattr(eng, "precision_ok") and eng.precision_ok(arch, F32_CLASS):
                        if not getattr(owner, "_warned_shared", False):
                            owner._warned_shared = True
                            log.warning("%s: another model's checkpoint on this engine keeps it out of the fp32-class mode; this model runs in "
                                        "exact fp32 while that one is loaded (%s)", what, e)
                        eng.set_precision(EXACT)
                    else:
                        raise
            return fn(eng)
        except ReidHipError as e:
            if owner._mode != F32_CLASS or not refused(e) or (e.status == -3 and (fault_before & 1)):
                raise
            if e.status == -3:
                eng.clear_fault()
            fallback_to_exact(owner, e, what)
        finally:
            if eng.precision != prev:
                try:
                    eng.set_precision(prev)
                except ReidHipError:      # `prev` was the fp32-class mode and the weights bound meanwhile cannot be split: stay
                    pass
    raise AssertionError("unreachable")
