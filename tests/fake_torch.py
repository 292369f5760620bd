"""TEST INFRASTRUCTURE: the small part of torch / torch.distributed that bench.py and ddt/engine.py touch, on numpy and threads -- so
that bench.py's whole N > 1 orchestration (communicator ids, the tree-sharded job, scaling_detail, every leg of other_modes incl. the
hybrid jobs, teardown order) can run with 2-8 ranks as THREADS against the CPU model of the host side (tests/mock_hip/).  "Device
memory" is host memory there, so a "cuda tensor" is a numpy array with a data_ptr; streams are the model's null stream per device."""
import threading
import types

import numpy as np

float32, float64, int32 = np.float32, np.float64, np.int32


class Device:
    type = "cuda"

    def __init__(self, index):
        self.index = index

    def __repr__(self):
        return f"cuda:{self.index}"


class Tensor:
    is_cuda = True

    def __init__(self, a, device):
        self.a, self.device = a, device

    dtype = property(lambda self: self.a.dtype)
    shape = property(lambda self: self.a.shape)

    def data_ptr(self):
        return self.a.ctypes.data

    def numel(self):
        return self.a.size

    def element_size(self):
        return self.a.itemsize

    def dim(self):
        return self.a.ndim

    def is_contiguous(self):
        return self.a.flags["C_CONTIGUOUS"]

    def __getitem__(self, k):
        return Tensor(self.a[k], self.device)

    def __len__(self):
        return len(self.a)

    def cpu(self):
        return self

    def numpy(self):
        return self.a

    def clone(self):
        return Tensor(self.a.copy(), self.device)

    def abs(self):
        return Tensor(np.abs(self.a), self.device)

    def max(self):
        return Tensor(np.asarray(self.a.max() if self.a.size else 0.0), self.device)

    def sum(self):
        return Tensor(np.asarray(self.a.sum()), self.device)

    def item(self):
        return self.a.reshape(-1)[0].item()

    def __sub__(self, o):
        return Tensor(self.a - o.a, self.device)

    def __ne__(self, o):
        return Tensor(self.a != o.a, self.device)


class World:
    """what the launcher + torch.distributed give N ranks: a barrier, an object broadcast, a MAX all-reduce"""

    def __init__(self, n):
        self.n, self.barrier, self.slots, self.local = n, threading.Barrier(n), {}, threading.local()


def make(world: World, mock_lib):
    """-> a module object that stands in for `torch` (with .cuda and .distributed) for the ranks (threads) of `world`"""
    t = types.ModuleType("torch")
    t.float32, t.float64, t.int32 = float32, float64, int32
    rank = lambda: world.local.rank

    def _dev(device):
        if isinstance(device, Device):
            return device
        if isinstance(device, str) and ":" in device:
            return Device(int(device.split(":")[1]))
        return Device(rank())

    t.device = lambda kind, index=0: Device(index)
    t.empty = lambda shape, dtype=float32, device=None: Tensor(np.full(shape, np.nan if np.issubdtype(dtype, np.floating) else -1, dtype), _dev(device))
    t.tensor = lambda data, dtype=float64, device=None: Tensor(np.asarray(data, dtype), _dev(device))
    t.from_numpy = lambda a: Tensor(a, Device(rank()))

    cuda = types.ModuleType("torch.cuda")
    cuda.is_available = lambda: True
    cuda.device_count = lambda: 8

    def set_device(i):
        assert mock_lib.hipSetDevice(int(i)) == 0

    def synchronize():
        assert mock_lib.hipDeviceSynchronize() == 0

    cuda.set_device, cuda.synchronize = set_device, synchronize
    cuda.current_stream = lambda device=None: types.SimpleNamespace(cuda_stream=None)   # the model's null stream of the thread's device
    t.cuda = cuda

    d = types.ModuleType("torch.distributed")
    d.ReduceOp = types.SimpleNamespace(MAX="max")
    d.init_process_group = lambda *a, **k: None
    d.destroy_process_group = lambda: None
    d.barrier = lambda: world.barrier.wait()

    def broadcast_object_list(box, src=0):
        if rank() == src:
            world.slots["bcast"] = list(box)
        world.barrier.wait()
        box[:] = world.slots["bcast"]
        world.barrier.wait()

    def all_reduce(tensor, op=None):
        world.slots[("ar", rank())] = tensor.a.copy()
        world.barrier.wait()
        m = np.maximum.reduce([world.slots[("ar", r)] for r in range(world.n)])
        world.barrier.wait()
        tensor.a[...] = m

    d.broadcast_object_list, d.all_reduce = broadcast_object_list, all_reduce
    t.distributed = d
    return t
