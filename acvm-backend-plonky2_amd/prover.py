"""ctypes binding of libp2gpu.so (include/p2gpu.h) with the reference's names.

Mirrors, for the hot path only:
  * ``CircuitData<F, C, D>``            -> :class:`CircuitData` (built from the circuit blob the
    Rust side exports once per circuit; see include/p2gpu.h)
  * ``circuit_data.prove(witnesses)``   -> :meth:`CircuitData.prove`
    (plonky2-backend/src/actions/prove_action.rs:96; test twin
    circuit_translation/tests/factories/utils.rs:26)
  * ``ProofWithPublicInputs::to_bytes`` -> :meth:`ProofWithPublicInputs.to_bytes`
    (prove_action.rs:75-78; compression stays on the Rust side)
Errors surface as :class:`P2GpuError` carrying the library's message, where the
reference would `unwrap()` an `anyhow::Error`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

UINT64_MAX = (1 << 64) - 1


class P2GpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"p2gpu error {code}: {msg}")
        self.code = code


class _Timings(ctypes.Structure):
    _fields_ = [
        ("wires_commit_ms", ctypes.c_double),
        ("zs_commit_ms", ctypes.c_double),
        ("quotient_ms", ctypes.c_double),
        ("openings_ms", ctypes.c_double),
        ("fri_ms", ctypes.c_double),
        ("total_ms", ctypes.c_double),
        ("h2d_ms", ctypes.c_double),
        ("pow_witness", ctypes.c_uint64),
    ]


_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64)


class _DevBytes:
    """Zero-copy view of a device buffer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def lib_path():
    # P2GPU_LIBRARY: an alternative build of the same library (A/B measurements of kernel variants only)
    return os.environ.get("P2GPU_LIBRARY") or os.path.join(_HERE, "libp2gpu.so")


def load_library():
    """Load libp2gpu.so.  Fails loudly when the HIP extension is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise P2GpuError(-8, f"{path} is missing: build it with __graft_entry__.build() (hipcc, gfx950)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  Importing torch FIRST makes libp2gpu.so bind to that already-loaded runtime,
    # so torch tensors (device memory, streams) and our kernels share one context; loading in
    # the other order maps two runtimes and the second one sees no devices.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is part of the target image
        pass
    lib = ctypes.CDLL(path)
    vp, sz, u8p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p
    lib.p2gpu_last_error.restype = ctypes.c_char_p
    lib.p2gpu_init.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.p2gpu_circuit_create.argtypes = [u8p, sz, ctypes.POINTER(vp)]
    lib.p2gpu_circuit_destroy.argtypes = [vp]
    lib.p2gpu_circuit_destroy.restype = None
    lib.p2gpu_circuit_cap.argtypes = [vp, u8p]
    lib.p2gpu_circuit_digest.argtypes = [vp, u8p]
    lib.p2gpu_proof_size_bound.argtypes = [vp]
    lib.p2gpu_proof_size_bound.restype = sz
    lib.p2gpu_circuit_device.argtypes = [vp]
    lib.p2gpu_circuit_hash_bytes.argtypes = [vp]
    lib.p2gpu_prove.argtypes = [vp, vp, vp, ctypes.c_uint32, u8p, ctypes.POINTER(sz), ctypes.POINTER(_Timings)]
    lib.p2gpu_prove_dev.argtypes = lib.p2gpu_prove.argtypes
    lib.p2gpu_prove_routed.argtypes = lib.p2gpu_prove.argtypes
    a = lib.p2gpu_prove.argtypes
    lib.p2gpu_prove_sparse.argtypes = [a[0], a[1], ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32] + list(a[2:])
    lib.p2gpu_fill_witness.argtypes = [vp, vp]
    lib.p2gpu_verify.argtypes = [vp, u8p, sz]
    lib.p2gpu_circuit_export_vk.argtypes = [vp, u8p, ctypes.POINTER(sz)]
    lib.p2gpu_circuit_export_vk_plonky2.argtypes = [vp, u8p, ctypes.POINTER(sz)]
    lib.p2gpu_verifier_create_plonky2.argtypes = [u8p, sz, ctypes.c_int, ctypes.POINTER(vp)]
    lib.p2gpu_verifier_create.argtypes = [u8p, sz, ctypes.POINTER(vp)]
    lib.p2gpu_proof_compress.argtypes = [vp, u8p, sz, u8p, ctypes.POINTER(sz)]
    lib.p2gpu_proof_decompress.argtypes = [vp, u8p, sz, u8p, ctypes.POINTER(sz)]
    lib.p2gpu_verify_compressed.argtypes = [vp, u8p, sz]
    lib.p2gpu_circuit_set.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint64]
    lib.p2gpu_circuit_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int, _ALLGATHER_FN, vp]
    lib.p2gpu_shard_unique_id.argtypes = [vp]
    lib.p2gpu_circuit_set_shard_rccl.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    lib.p2gpu_kernel_stats.argtypes = [vp, ctypes.c_char_p, vp, vp, vp, ctypes.c_int]
    lib.p2gpu_ifft_batch.argtypes = [vp, sz, ctypes.c_uint, vp]
    lib.p2gpu_lde_batch.argtypes = [vp, sz, ctypes.c_uint, ctypes.c_uint, vp]
    lib.p2gpu_commit_values.argtypes = [vp, sz, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, u8p]
    lib.p2gpu_hash_rows.argtypes = [vp, sz, sz, u8p]
    lib.p2gpu_field_selftest.argtypes = [vp, vp, sz, vp]
    lib.p2gpu_field_selftest16.argtypes = [vp, vp, sz, vp]
    lib.p2gpu_device_info.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(sz)]
    lib.p2gpu_peer_access.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.p2gpu_circuit_create_on.argtypes = [u8p, sz, ctypes.c_int, ctypes.POINTER(vp)]
    lib.p2gpu_peer_access.restype = ctypes.c_int
    lib.p2gpu_host_alloc.argtypes = [sz]
    lib.p2gpu_host_alloc.restype = vp
    lib.p2gpu_host_free.argtypes = [vp]
    lib.p2gpu_host_free.restype = None
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise P2GpuError(rc, load_library().p2gpu_last_error().decode(errors="replace"))


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def init(devices=(0,)):
    """``p2gpu_init``: the HIP devices this process drives.  One id: one process per GPU.  Several ids: every
    ``CircuitData`` made afterwards is a device group and each ``prove`` is ONE proof coset-sharded over those GPUs from
    this single process (peer-to-peer exchanges between the ranks' streams, one host thread per rank inside the call) --
    the shape the reference's one ``circuit_data.prove`` call site (prove_action.rs:96) can drive.  The number of ids
    must divide the 8 LDE cosets; an id may repeat (ranks sharing a GPU: the functional test of this path on one GPU)."""
    lib = load_library()
    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    _check(lib.p2gpu_init(devs, len(devices)))


class _HostBlock:
    """Owner of one ``p2gpu_host_alloc`` block; the numpy view made over it keeps this object alive."""

    def __init__(self, nbytes):
        self._lib = load_library()
        self.ptr = self._lib.p2gpu_host_alloc(nbytes)
        if not self.ptr:
            raise P2GpuError(-3, self._lib.p2gpu_last_error().decode(errors="replace"))
        self.buf = (ctypes.c_uint8 * max(1, nbytes)).from_address(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            self.buf = None
            self._lib.p2gpu_host_free(self.ptr)
            self.ptr = None


def host_array(shape, dtype=np.uint64):
    """A numpy array in page-locked host memory (``p2gpu_host_alloc``): a wire matrix built in place here uploads by
    direct DMA inside ``prove`` instead of being staged by the HIP runtime on the calling thread."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    blk = _HostBlock(nbytes)
    arr = np.frombuffer(blk.buf, dtype=dtype, count=nbytes // np.dtype(dtype).itemsize).reshape(shape)
    _HOST_BLOCKS[id(blk)] = blk  # freed by host_free(arr) or at interpreter exit; a view may outlive `arr` itself
    arr.flags.writeable = True
    return arr


_HOST_BLOCKS = {}


def host_free(arr):
    """Release the block behind an array returned by ``host_array`` (the array must not be used afterwards)."""
    addr = arr.ctypes.data
    for k, blk in list(_HOST_BLOCKS.items()):
        if blk.ptr == addr:
            del _HOST_BLOCKS[k]
            blk.__del__()
            return
    raise P2GpuError(-1, "host_free: not an array from host_array")


def device_info():
    lib = load_library()
    name = ctypes.create_string_buffer(256)
    cu = ctypes.c_int()
    mem = ctypes.c_size_t()
    _check(lib.p2gpu_device_info(name, 256, ctypes.byref(cu), ctypes.byref(mem)))
    return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}


def peer_access():
    """n x n matrix for the device ids of the last ``init``: 1 = device a reaches device b's memory directly (peer copies
    over xGMI), 0 = it does not (staged through the host), -1 = the same device (``p2gpu_peer_access``)."""
    lib = load_library()
    n = lib.p2gpu_peer_access(None, 0)
    if n < 0:
        _check(n)
    m = (ctypes.c_int * (n * n))()
    _check(min(0, lib.p2gpu_peer_access(m, n * n)))
    return [[int(m[a * n + b]) for b in range(n)] for a in range(n)]


class ProofWithPublicInputs:
    """Uncompressed proof bytes in plonky2's ``ProofWithPublicInputs::to_bytes`` layout."""

    def __init__(self, data, timings=None):
        self._data = bytes(data)
        self.timings = timings or {}

    def to_bytes(self):
        return self._data

    def __len__(self):
        return len(self._data)

    def __eq__(self, other):
        return isinstance(other, ProofWithPublicInputs) and self._data == other._data


class CircuitData:
    """Prover-side circuit handle (device-resident constants/sigmas oracle, root tables)."""

    def __init__(self, blob, device=None):
        """device: None = what ``init`` selected (several ids: a device group, every proof sharded over them); an id of that
        list = a plain handle on that one device (``p2gpu_circuit_create_on``: replicas across GPUs from one process)."""
        lib = load_library()
        self._lib = lib
        self._blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        hdr = self._blob[:256].view(np.uint32)
        self.degree_bits = int(hdr[2])
        self.num_wires = int(hdr[3])
        self.num_routed_wires = int(hdr[4])
        self.rate_bits = int(hdr[9])
        self.cap_height = int(hdr[10])
        self.num_public_inputs = int(hdr[24])
        self._h = ctypes.c_void_p()
        if device is None:
            _check(lib.p2gpu_circuit_create(self._blob.ctypes.data, self._blob.nbytes, ctypes.byref(self._h)))
        else:
            _check(lib.p2gpu_circuit_create_on(self._blob.ctypes.data, self._blob.nbytes, int(device), ctypes.byref(self._h)))
        self._bound = lib.p2gpu_proof_size_bound(self._h)

    @classmethod
    def from_blob(cls, blob):
        return cls(blob)

    @property
    def degree(self):
        return 1 << self.degree_bits

    def constants_sigmas_cap(self):
        out = np.zeros(self.hash_bytes() << self.cap_height, dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_cap(self._h, out.ctypes.data))
        return out.tobytes()

    def circuit_digest(self):
        out = np.zeros(self.hash_bytes(), dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_digest(self._h, out.ctypes.data))
        return out.tobytes()

    def hash_bytes(self):
        """Digest size of the circuit's hasher: 25 (KeccakHash<25>) or 32 (PoseidonHash)."""
        return int(self._lib.p2gpu_circuit_hash_bytes(self._h))

    def device_index(self):
        """HIP device the handle's buffers and streams live on."""
        return int(self._lib.p2gpu_circuit_device(self._h))

    def set(self, key, value):
        _check(self._lib.p2gpu_circuit_set(self._h, key.encode(), ctypes.c_uint64(value)))

    def set_shard(self, rank, world, group=None, transport=None):
        """Coset-shard every following proof over the `world` ranks of `group` (one process per GPU;
        torch.distributed must be initialised when world > 1).

        transport "rccl" (default whenever the process group's backend is nccl): the library itself
        calls RCCL (ncclAllGather on its own HIP stream, xGMI on a real node) -- torch.distributed is used
        once, to broadcast the 128-byte ncclUniqueId.  transport "callback" (default for gloo, i.e. the
        CPU-side tests): the library hands device buffers to a host callback that gathers them with
        torch.distributed, staging through host tensors."""
        import torch
        import torch.distributed as dist

        backend = dist.get_backend(group) if (world > 1 or (dist.is_available() and dist.is_initialized())) else None
        if transport is None:
            transport = "rccl" if backend == "nccl" else "callback"
        if transport == "rccl":
            idbuf = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                _check(self._lib.p2gpu_shard_unique_id(idbuf.ctypes.data))
            if world > 1:
                if backend == "nccl":
                    t = torch.from_numpy(idbuf).cuda()
                    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    idbuf = t.cpu().numpy()
                else:
                    t = torch.from_numpy(idbuf)
                    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    idbuf = t.numpy()
            idbuf = np.ascontiguousarray(idbuf)
            _check(self._lib.p2gpu_circuit_set_shard_rccl(self._h, rank, world, idbuf.ctypes.data))
            self._shard_cb = None
            self.shard = (rank, world)
            return
        if world > 1:

            def _allgather(_ctx, send, recv, nbytes):
                try:
                    s = torch.as_tensor(_DevBytes(send, nbytes), device="cuda")
                    r = torch.as_tensor(_DevBytes(recv, nbytes * world), device="cuda")
                    if backend == "nccl":
                        dist.all_gather_into_tensor(r, s, group=group)
                    else:
                        out = torch.empty(nbytes * world, dtype=torch.uint8)
                        dist.all_gather_into_tensor(out, s.cpu(), group=group)
                        r.copy_(out)
                    torch.cuda.synchronize()
                    return 0
                except Exception as e:  # never let an exception cross the C ABI
                    self._shard_error = e
                    return -1

            self._shard_cb = _ALLGATHER_FN(_allgather)  # keep alive as long as the handle
        else:
            self._shard_cb = _ALLGATHER_FN(0)
        _check(self._lib.p2gpu_circuit_set_shard(self._h, rank, world, self._shard_cb, None))
        self.shard = (rank, world)

    def kernel_stats(self):
        cap = 64
        names = ctypes.create_string_buffer(64 * cap)
        ms = np.zeros(cap, dtype=np.float64)
        by = np.zeros(cap, dtype=np.float64)
        ln = np.zeros(cap, dtype=np.uint64)
        k = self._lib.p2gpu_kernel_stats(self._h, names, ms.ctypes.data, by.ctypes.data, ln.ctypes.data, cap)
        out = {}
        for i in range(max(k, 0)):
            nm = names.raw[64 * i:64 * (i + 1)].split(b"\0", 1)[0].decode()
            out[nm] = {"ms": float(ms[i]), "bytes": float(by[i]), "launches": int(ln[i])}
        return out

    def prove(self, wires, public_inputs=()):
        """``circuit_data.prove(witnesses)``: `wires` is the full witness matrix
        [num_wires][degree] (numpy uint64 on the host, or a torch tensor on the
        circuit's GPU, which is read in place)."""
        pis = _u64(np.array(list(public_inputs), dtype=np.uint64))
        out = np.zeros(self._bound, dtype=np.uint8)
        plen = ctypes.c_size_t(out.nbytes)
        tm = _Timings()
        expect = self.num_wires * self.degree
        if hasattr(wires, "data_ptr") and getattr(wires, "is_cuda", False):
            if wires.numel() != expect or wires.element_size() != 8 or not wires.is_contiguous():
                raise P2GpuError(-7, "wires tensor must be contiguous int64/uint64 [num_wires][degree]")
            if wires.device.index != self.device_index():
                raise P2GpuError(-7, f"wires tensor lives on cuda:{wires.device.index}, the circuit on cuda:{self.device_index()}")
            rc = self._lib.p2gpu_prove_dev(self._h, ctypes.c_void_p(wires.data_ptr()), pis.ctypes.data, len(pis),
                                           out.ctypes.data, ctypes.byref(plen), ctypes.byref(tm))
        else:
            w = _u64(wires)
            if w.size != expect:
                raise P2GpuError(-7, f"wires must be [num_wires={self.num_wires}][degree={self.degree}]")
            rc = self._lib.p2gpu_prove(self._h, w.ctypes.data, pis.ctypes.data, len(pis), out.ctypes.data,
                                       ctypes.byref(plen), ctypes.byref(tm))
        _check(rc)
        t = {f: getattr(tm, f) for f, _ in _Timings._fields_}
        return ProofWithPublicInputs(out[:plen.value].tobytes(), t)

    def fill_witness(self, wires_dev):
        """Run the gates' row-local generators on a device wire matrix (torch tensor, in place)."""
        if not (hasattr(wires_dev, "data_ptr") and getattr(wires_dev, "is_cuda", False)):
            raise P2GpuError(-7, "fill_witness needs a torch tensor on the GPU")
        if wires_dev.numel() != self.num_wires * self.degree or not wires_dev.is_contiguous():
            raise P2GpuError(-7, "wires tensor must be contiguous [num_wires][degree]")
        _check(self._lib.p2gpu_fill_witness(self._h, ctypes.c_void_p(wires_dev.data_ptr())))
        return wires_dev

    def prove_routed(self, routed, public_inputs=()):
        """Prove from the routed columns only ([num_routed_wires][degree], host): gate-internal
        columns are derived on the GPU by the row-local generators."""
        pis = _u64(np.array(list(public_inputs), dtype=np.uint64))
        r = _u64(routed)
        if r.size != self.num_routed_wires * self.degree:
            raise P2GpuError(-7, f"routed must be [num_routed_wires={self.num_routed_wires}][degree={self.degree}]")
        out = np.zeros(self._bound, dtype=np.uint8)
        plen = ctypes.c_size_t(out.nbytes)
        tm = _Timings()
        _check(self._lib.p2gpu_prove_routed(self._h, r.ctypes.data, pis.ctypes.data, len(pis), out.ctypes.data,
                                            ctypes.byref(plen), ctypes.byref(tm)))
        return ProofWithPublicInputs(out[:plen.value].tobytes(), {f: getattr(tm, f) for f, _ in _Timings._fields_})

    def prove_sparse(self, wires, ncols, row, public_inputs=(), tail=None):
        """Prove from the first `ncols` wire columns plus ONE value for each of the others (``p2gpu_prove_sparse``):
        column j >= ncols is zero in every row but `row`, where it holds ``tail[j - ncols]`` -- what plonky2 leaves
        in the wires no gate of a circuit uses.  `wires` is the full [num_wires][degree] matrix (the tail is then
        read off row `row`) or just its first `ncols` columns with `tail` given.  Same bytes as ``prove``; only
        `ncols` columns cross PCIe."""
        pis = _u64(np.array(list(public_inputs), dtype=np.uint64))
        W, n = self.num_wires, self.degree
        w = _u64(wires)
        if not (0 <= ncols <= W and 0 <= row < n):
            raise P2GpuError(-7, f"prove_sparse: ncols {ncols} of {W} wires, row {row} of {n}")
        if tail is None:
            if w.size != W * n:
                raise P2GpuError(-7, f"wires must be [num_wires={W}][degree={n}] when no tail is given")
            full = w.reshape(W, n)
            tail = np.ascontiguousarray(full[ncols:, row])
            w = np.ascontiguousarray(full[:ncols]).reshape(-1)
        else:
            tail = _u64(np.asarray(tail, dtype=np.uint64))
        if ncols > W or w.size != ncols * n or tail.size != W - ncols:
            raise P2GpuError(-7, f"prove_sparse: [{ncols}][{n}] dense columns and {W - ncols} tail values expected")
        out = np.zeros(self._bound, dtype=np.uint8)
        plen = ctypes.c_size_t(out.nbytes)
        tm = _Timings()
        _check(self._lib.p2gpu_prove_sparse(self._h, w.ctypes.data if w.size else None, ncols,
                                            tail.ctypes.data if tail.size else None, row,
                                            pis.ctypes.data, len(pis), out.ctypes.data, ctypes.byref(plen), ctypes.byref(tm)))
        return ProofWithPublicInputs(out[:plen.value].tobytes(), {f: getattr(tm, f) for f, _ in _Timings._fields_})

    def verify(self, proof):
        """``circuit_data.verify(proof)``: raises P2GpuError(P2GPU_E_VERIFY) when the proof is
        rejected (the failed check is the message), returns None when it is accepted."""
        _verify(self._lib, self._h, proof)

    def verifier_data(self):
        """``circuit_data.verifier_data()``: the verifier's share of the circuit."""
        return VerifierCircuitData(self.verifier_blob())

    def verifier_blob(self):
        """Bytes of the verifier's share (header, gate table, constants_sigmas cap, digest, k_is)."""
        n = ctypes.c_size_t(0)
        _check(self._lib.p2gpu_circuit_export_vk(self._h, None, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_export_vk(self._h, out.ctypes.data, ctypes.byref(n)))
        return out[:n.value].tobytes()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.p2gpu_circuit_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _verify(lib, handle, proof):
    data = proof.to_bytes() if hasattr(proof, "to_bytes") else bytes(proof)
    buf = np.frombuffer(data, dtype=np.uint8)
    _check(lib.p2gpu_verify(handle, buf.ctypes.data if buf.size else None, buf.size))


def _convert(fn, handle, data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = ctypes.c_size_t(0)
    _check(fn(handle, buf.ctypes.data if buf.size else None, buf.size, None, ctypes.byref(n)))
    out = np.zeros(max(n.value, 1), dtype=np.uint8)
    _check(fn(handle, buf.ctypes.data if buf.size else None, buf.size, out.ctypes.data, ctypes.byref(n)))
    return out[:n.value].tobytes()


class _ProofFormats:
    """`proof.compress(..)` / `.decompress(..)` / `verify_compressed` of the reference's prove and
    verify actions (prove_action.rs:75-78, verify_action.rs:11-17), on either kind of handle."""

    def compress(self, proof):
        data = proof.to_bytes() if hasattr(proof, "to_bytes") else bytes(proof)
        return _convert(self._lib.p2gpu_proof_compress, self._h, data)

    def decompress(self, compressed):
        return ProofWithPublicInputs(_convert(self._lib.p2gpu_proof_decompress, self._h, compressed))

    def verify_compressed(self, compressed):
        buf = np.frombuffer(bytes(compressed), dtype=np.uint8)
        _check(self._lib.p2gpu_verify_compressed(self._h, buf.ctypes.data if buf.size else None, buf.size))

    def to_plonky2_bytes(self):
        """``verifier_data.to_bytes(&BackendGateSerializer)``: the VK file `write_vk` stores
        (write_vk_action.rs:77-80).  UNPINNED restatement of plonky2's serialization (include/p2gpu.h)."""
        n = ctypes.c_size_t(0)
        _check(self._lib.p2gpu_circuit_export_vk_plonky2(self._h, None, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_export_vk_plonky2(self._h, out.ctypes.data, ctypes.byref(n)))
        return out[:n.value].tobytes()


# the prover handle offers the same three conversions
CircuitData.compress = _ProofFormats.compress
CircuitData.decompress = _ProofFormats.decompress
CircuitData.verify_compressed = _ProofFormats.verify_compressed
CircuitData.to_plonky2_vk_bytes = _ProofFormats.to_plonky2_bytes


class VerifierCircuitData(_ProofFormats):
    """Verifier-only handle (``VerifierCircuitData`` of the reference's verify action,
    plonky2-backend/src/actions/verify_action.rs:11-17).  Host code only: works without a GPU."""

    def __init__(self, blob):
        lib = load_library()
        self._lib = lib
        self._blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        self._h = ctypes.c_void_p()
        _check(lib.p2gpu_verifier_create(self._blob.ctypes.data, self._blob.nbytes, ctypes.byref(self._h)))
        self._read_header()

    @classmethod
    def from_plonky2_bytes(cls, vk, hasher=0):
        """``VerifierCircuitData::from_bytes(buffer, &BackendGateSerializer)`` of the reference's verify action
        (noir_and_plonky2_serialization.rs:16-22).  UNPINNED restatement of plonky2's serialization."""
        self = cls.__new__(cls)
        lib = load_library()
        self._lib = lib
        buf = np.ascontiguousarray(np.frombuffer(bytes(vk), dtype=np.uint8))
        self._h = ctypes.c_void_p()
        _check(lib.p2gpu_verifier_create_plonky2(buf.ctypes.data if buf.size else None, buf.size, int(hasher), ctypes.byref(self._h)))
        n = ctypes.c_size_t(0)
        _check(lib.p2gpu_circuit_export_vk(self._h, None, ctypes.byref(n)))
        self._blob = np.zeros(n.value, dtype=np.uint8)
        _check(lib.p2gpu_circuit_export_vk(self._h, self._blob.ctypes.data, ctypes.byref(n)))
        self._read_header()
        return self

    def _read_header(self):
        hdr = self._blob[:256].view(np.uint32)
        self.degree_bits = int(hdr[2])
        self.cap_height = int(hdr[10])
        self.num_public_inputs = int(hdr[24])

    def to_bytes(self):
        return self._blob.tobytes()

    def hash_bytes(self):
        return int(self._lib.p2gpu_circuit_hash_bytes(self._h))

    def circuit_digest(self):
        out = np.zeros(self.hash_bytes(), dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_digest(self._h, out.ctypes.data))
        return out.tobytes()

    def constants_sigmas_cap(self):
        out = np.zeros(self.hash_bytes() << self.cap_height, dtype=np.uint8)
        _check(self._lib.p2gpu_circuit_cap(self._h, out.ctypes.data))
        return out.tobytes()

    def verify(self, proof):
        _verify(self._lib, self._h, proof)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.p2gpu_circuit_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- build(): the GPU-free precompute + circuit creation -----------------------
class _BuildParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("degree_bits", "num_wires", "num_routed_wires", "num_challenges", "quotient_degree_factor",
                                                "rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds", "num_public_inputs")]


class _GateDecl(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("p", ctypes.c_uint32 * 4), ("degree", ctypes.c_uint32), ("num_constants", ctypes.c_uint32)]


def build_blob(degree_bits, gates, row_gate, row_constants, copies, num_public_inputs=0, num_wires=234, num_routed_wires=80,
               num_challenges=2, quotient_degree_factor=8, rate_bits=3, cap_height=4, proof_of_work_bits=16, num_query_rounds=28):
    """``builder.build::<C>()`` minus the GPU part (circuit_translation/mod.rs:80-82): selector columns and groups,
    sigma polynomials from the copy constraints, k_is, FRI arities -> circuit blob (p2gpu_build_blob).
    gates: [(kind, (p0, p1, p2, p3), degree, num_constants)] sorted by (degree, id) like CommonCircuitData.gates;
    row_gate[n]; row_constants[max num_constants][n]; copies[m][4] = (row_a, col_a, row_b, col_b).
    ``CircuitData(build_blob(...))`` then commits constants and sigmas on the GPU and derives the digest."""
    lib = load_library()
    lib.p2gpu_build_blob.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    bp = _BuildParams(degree_bits, num_wires, num_routed_wires, num_challenges, quotient_degree_factor, rate_bits, cap_height,
                      proof_of_work_bits, num_query_rounds, num_public_inputs)
    gd = (_GateDecl * len(gates))()
    for i, (kind, ps, deg, nk) in enumerate(gates):
        gd[i].kind, gd[i].degree, gd[i].num_constants = kind, deg, nk
        for j, v in enumerate(tuple(ps) + (0,) * (4 - len(ps))):
            gd[i].p[j] = v
    rg = np.ascontiguousarray(row_gate, dtype=np.uint32)
    rc = np.ascontiguousarray(row_constants, dtype=np.uint64)
    cp = np.ascontiguousarray(copies, dtype=np.uint32).reshape(-1, 4)
    if rg.size != 1 << degree_bits:
        raise P2GpuError(-7, "row_gate must have 2^degree_bits entries")
    ln = ctypes.c_size_t(0)
    args = [ctypes.byref(bp), gd, len(gates), rg.ctypes.data, rc.ctypes.data if rc.size else None, cp.ctypes.data if cp.size else None, len(cp)]
    _check(lib.p2gpu_build_blob(*args, None, ctypes.byref(ln)))
    out = np.zeros(ln.value, dtype=np.uint8)
    _check(lib.p2gpu_build_blob(*args, out.ctypes.data, ctypes.byref(ln)))
    return out[:ln.value]


# ---- stage-level operators ---------------------------------------------------
def ifft_batch(vals):
    """PolynomialValues::ifft on every row of `vals` [ncols][2^d]."""
    lib = load_library()
    v = _u64(vals)
    ncols, n = v.shape
    d = n.bit_length() - 1
    out = np.zeros_like(v)
    _check(lib.p2gpu_ifft_batch(v.ctypes.data, ncols, d, out.ctypes.data))
    return out


def lde_batch(coeffs, rate_bits=3):
    """PolynomialCoeffs::lde(rate_bits).coset_fft(7) on every row of `coeffs`."""
    lib = load_library()
    c = _u64(coeffs)
    ncols, n = c.shape
    d = n.bit_length() - 1
    out = np.zeros((ncols, n << rate_bits), dtype=np.uint64)
    _check(lib.p2gpu_lde_batch(c.ctypes.data, ncols, d, rate_bits, out.ctypes.data))
    return out


def commit_values(vals, rate_bits=3, cap_height=4):
    """PolynomialBatch::from_values(...).merkle_tree.cap as bytes (2^cap_height x 25)."""
    lib = load_library()
    v = _u64(vals)
    ncols, n = v.shape
    d = n.bit_length() - 1
    out = np.zeros(25 << cap_height, dtype=np.uint8)
    _check(lib.p2gpu_commit_values(v.ctypes.data, ncols, d, rate_bits, cap_height, out.ctypes.data))
    return out.tobytes()


def field_selftest(a, b):
    """Mismatch counters of the device field primitives against the portable code on the word pairs (a[i], b[i])."""
    lib = load_library()
    x = np.ascontiguousarray(a, dtype=np.uint64)
    y = np.ascontiguousarray(b, dtype=np.uint64)
    assert x.shape == y.shape and x.ndim == 1
    bad = np.zeros(16, dtype=np.uint64)
    _check(lib.p2gpu_field_selftest16(x.ctypes.data, y.ctypes.data, x.size, bad.ctypes.data))
    return bad


def hash_rows(rows):
    """KeccakHash<25>::hash_or_noop of every row."""
    lib = load_library()
    r = _u64(rows)
    n_rows, row_len = r.shape
    out = np.zeros(25 * n_rows, dtype=np.uint8)
    _check(lib.p2gpu_hash_rows(r.ctypes.data, n_rows, row_len, out.ctypes.data))
    return out.reshape(n_rows, 25)
