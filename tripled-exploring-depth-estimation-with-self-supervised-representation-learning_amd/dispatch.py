"""Book-keeping of which implementation ran: hand-written HIP entry points vs ATen compositions.

Every autograd wrapper in ``ops.py`` reports its C-ABI calls through ``hip(name)``; every call site in
``mono/model`` that can route a HIP-resident tensor to an ATen composition instead (a layout the kernels do
not take: not channels_last, channel count not a multiple of 8/64, ...) reports it through
``fallback(site, why)``.  In strict mode a fallback raises, so a layout regression cannot silently move a
training step off the hand-written kernels (``bench.py`` runs strict and prints ``"fallbacks": 0``).
"""
import collections
import contextlib
import logging
import os


class FallbackError(RuntimeError):
    """A HIP-resident tensor was about to take an ATen composition although strict mode is on."""


_strict = [os.environ.get("TD_STRICT", "0") == "1"]
hip_calls = collections.Counter()
fallbacks = collections.Counter()


_trace = os.environ.get("TD_TRACE_CALLS", "0") == "1"


def hip(name, n=1):
    hip_calls[name] += n
    if _trace:      # fault triage: with AMD_SERIALIZE_KERNEL=3 the last line printed names the entry point that faulted
        import sys
        sys.stderr.write("td-call %s\n" % name)
        sys.stderr.flush()


def fallback(site, why=""):
    if not fallbacks[site]:      # once per site: a training run that is off the hand-written kernels says so in its log
        logging.getLogger("tripled_amd").warning("%s: HIP tensor routed to ATen ops instead of the hand-written kernel%s",
                                                 site, (" (" + why + ")") if why else "")
    fallbacks[site] += 1
    if _strict[0]:
        raise FallbackError("%s fell back to ATen ops%s (strict mode)" % (site, (": " + why) if why else ""))


def set_strict(on):
    prev = _strict[0]
    _strict[0] = bool(on)
    return prev


@contextlib.contextmanager
def strict(on=True):
    prev = set_strict(on)
    try:
        yield
    finally:
        set_strict(prev)


def reset():
    hip_calls.clear()
    fallbacks.clear()


def snapshot():
    return {"hip_calls": dict(hip_calls), "fallbacks": dict(fallbacks)}
