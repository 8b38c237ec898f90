"""One set of side streams per device, by role.

The runtime binds a stream to one of GPU_MAX_HW_QUEUES hardware queues (8, set by the package; the default is 4) when it is first
used, round robin, and launches of two streams that share a queue serialise.  Components that each create their own streams
(an encoder's three backward lanes, a trainer's encoder stream, the inference lanes, ...) make that binding depend on how many
streams the process has created before: the cnn_finetune step of a trainer built after a dozen others ran at 1.9k images/s
instead of 3.4k (bf16x3 plan, bench.py extras) with its weight-gradient lane on the main stream's queue.  Every component
therefore takes its lanes from this table: SEVEN streams per device for the life of the process (one hardware queue each beside
the default stream's), created together in a fixed order.

Rules of use:
* A role is ONE stream, shared by every component that asks for it: two live users of a role (two trainers' 'wgrad' lanes, the
  train and the eval model's 'encoder') are ordered one behind the other -- correct, never concurrent -- and a hipGraph capture
  that forks onto a role's stream puts that stream into capture mode for its other users too: components that share a role
  must not run or be captured at the same time (the harnesses run train / eval / inference stages one after the other).
* The inference lanes 2..4 (`CaptionModel.infer_pipelined` with more than two batches in flight) are ALIASES of the training-only
  lanes 'wgrad', 'chain1', 'aux': decode loops and a cnn_finetune backward never run at once, and ten streams on eight queues
  would bring the history-dependent binding back.
* Only the 'encoder' role has a priority variant (COMIC_SIDE_PRIORITY, an experiment switch): one extra stream, created behind
  the seven, so it cannot shift their binding."""

STREAMS = ('encoder', 'wgrad', 'chain1', 'aux', 'infer0', 'infer1', 'comm')
ALIASES = {'infer2': 'wgrad', 'infer3': 'chain1', 'infer4': 'aux'}
ROLES = STREAMS + tuple(ALIASES)
_LANES = {}


def lane(torch, device, role, priority=0):
    """The stream of `role` on `device` (created with its siblings on first use)."""
    assert role in ROLES, role
    role = ALIASES.get(role, role)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _LANES:
        _LANES[idx] = {r: torch.cuda.Stream(device=dev) for r in STREAMS}
    if int(priority) != 0:
        assert role == 'encoder', 'only the encoder lane has a priority variant'
        key = ('encoder', int(priority))
        if key not in _LANES[idx]:
            _LANES[idx][key] = torch.cuda.Stream(device=dev, priority=int(priority))
        return _LANES[idx][key]
    return _LANES[idx][role]
