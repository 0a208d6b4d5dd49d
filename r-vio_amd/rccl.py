"""Direct RCCL binding (ctypes) for the one collective of the sharded updater.

`torch.distributed` is used for the rendezvous only (it broadcasts the RCCL unique id).  The all-gather itself is issued
with ncclAllGather ON THE HANDLE'S FILTER STREAM: it is ordered behind the kernel that produced the block and ahead of the
kernel that consumes the gathered blocks by plain stream order — no helper stream, no cross-stream events, no tensor
wrappers per frame (the process-group path costs ~170 us per frame in exactly those).
"""
import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat64 = 8          # ncclDataType_t (nccl.h): ncclDouble = ncclFloat64 = 8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _lib(torch):
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    L = C.CDLL(path)
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    return L


class RcclComm:
    def __init__(self, rank, world, dist, torch):
        self.L = _lib(torch)
        uid = _UniqueId()
        if rank == 0:
            self._ck(self.L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES) if rank == 0 else None]   # (.internal would stop at the first NUL)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        self._ck(self.L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        self.world = world

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.L.ncclGetErrorString(rc).decode()))

    def all_gather_f64(self, send_ptr, recv_ptr, count, stream_ptr):
        """recv[r*count:(r+1)*count] = rank r's send[0:count] (doubles), enqueued on the given hipStream_t"""
        self._ck(self.L.ncclAllGather(C.c_void_p(send_ptr), C.c_void_p(recv_ptr), count, ncclFloat64, self.comm, C.c_void_p(stream_ptr)), "ncclAllGather")

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
