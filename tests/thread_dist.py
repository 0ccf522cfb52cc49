"""In-process stand-in for torch.distributed: N virtual ranks as threads of one process (one GPU).

RCCL refuses two ranks on the same device, so the multi-rank HIP path is exercised on a single GPU by running
point_cloud_viewer_amd.distributed unchanged against this object: same calls (all_reduce, all_gather,
batch_isend_irecv with P2POp, gather_object), executed by rendezvous on shared memory. Test infrastructure only."""
import threading
from types import SimpleNamespace

import torch


class ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.mail = {}
        self.lock = threading.Condition()

    def dist(self, rank):
        return ThreadDist(self, rank)


class _Done:
    def wait(self):
        return True


class ThreadDist:
    ReduceOp = SimpleNamespace(SUM="sum", MIN="min", MAX="max")

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.w.world

    def barrier(self):
        self.w.barrier.wait()

    def _exchange(self, value):
        self.w.slots[self.rank] = value
        self.w.barrier.wait()
        vals = list(self.w.slots)
        self.w.barrier.wait()
        return vals

    def all_reduce(self, t, op="sum"):
        torch.cuda.synchronize() if t.is_cuda else None
        vals = torch.stack(self._exchange(t.clone()))
        if op == "sum":
            red = vals.sum(dim=0).to(t.dtype)
        elif op == "min":
            red = vals.min(dim=0).values
        else:
            red = vals.max(dim=0).values
        t.copy_(red)

    def all_gather(self, out, t):
        for o, v in zip(out, self._exchange(t.clone())):
            o.copy_(v)

    # point-to-point: a batch posts every send, then matches receives in posting order per (src, dst)
    isend = "isend"
    irecv = "irecv"

    @staticmethod
    def P2POp(op, tensor, peer):
        return SimpleNamespace(op=op, tensor=tensor, peer=peer)

    def batch_isend_irecv(self, ops):
        """Point-to-point like the real thing: only the ranks named in the ops take part (no world-wide rendezvous)."""
        with self.w.lock:
            for o in ops:
                if o.op == "isend":
                    self.w.mail.setdefault((self.rank, o.peer), []).append(o.tensor.clone())
            self.w.lock.notify_all()
        for o in ops:
            if o.op == "irecv":
                with self.w.lock:
                    ok = self.w.lock.wait_for(lambda: self.w.mail.get((o.peer, self.rank)), timeout=120)
                    if not ok:
                        raise TimeoutError(f"rank {self.rank}: no message from rank {o.peer}")
                    msg = self.w.mail[(o.peer, self.rank)].pop(0)
                o.tensor.copy_(msg)
        return [_Done() for _ in ops]

    def gather_object(self, obj, out, dst=0):
        vals = self._exchange(obj)
        if self.rank == dst:
            out[:] = vals


def run_ranks(world, fn):
    """Run fn(rank, dist) on `world` threads; re-raise the first failure."""
    tw = ThreadWorld(world)
    errors = [None] * world
    results = [None] * world

    def body(r):
        try:
            results[r] = fn(r, tw.dist(r))
        except BaseException as e:  # noqa: BLE001 - surfaced below
            errors[r] = e
            tw.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results
