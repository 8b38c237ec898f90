"""One set of side streams per device, by role.

The runtime binds a stream to one of GPU_MAX_HW_QUEUES hardware queues (8, set by the package; the default is 4) when it is first
used, round robin, and launches of two streams that share a queue serialise.  Components that each create their own streams
(an encoder's three backward lanes, a trainer's encoder stream, the inference lanes, ...) make that binding depend on how many
streams the process has created before: the cnn_finetune step of a trainer built after a dozen others ran at 1.9k images/s
instead of 3.4k (bf16x3 plan, bench.py extras) with its weight-gradient lane on the main stream's queue.  Every component
therefore takes its lanes from this table: a role is ONE stream per device for the life of the process, the roles are created
in a fixed order, and there are no more of them than hardware queues beside the default stream.  Sharing a stream between
components only adds ordering; the roles that run at the same time inside one step are distinct."""

ROLES = ('encoder', 'wgrad', 'chain1', 'aux', 'infer0', 'infer1', 'comm', 'infer2', 'infer3', 'infer4')
_LANES = {}


def lane(torch, device, role, priority=0):
    """The stream of `role` on `device` (created with its siblings on first use)."""
    assert role in ROLES, role
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(priority))
    if key not in _LANES:
        _LANES[key] = {r: torch.cuda.Stream(device=dev, priority=int(priority)) for r in ROLES}
    return _LANES[key][role]
