"""acvm-backend-plonky2_amd -- MI355X-native `prove` hot path for eryxcoop/acvm-backend-plonky2.

Host-side mirror of the one reference interface on the path,
``CircuitData::prove(PartialWitness) -> ProofWithPublicInputs``
(plonky2-backend/src/actions/prove_action.rs:91-97), on top of the C ABI in
include/p2gpu.h (libp2gpu.so, hand-written HIP for gfx950).  There is no CPU
fallback: without the HIP library or without a GPU every call raises.

The directory name carries a hyphen (it is the repo's package name); import it
with ``__graft_entry__.load_package()`` which registers it as
``acvm_backend_plonky2_amd``.
"""
from .prover import (  # noqa: F401
    CircuitData,
    build_blob,
    VerifierCircuitData,
    ProofWithPublicInputs,
    P2GpuError,
    device_info,
    peer_access,
    init,
    host_array,
    host_free,
    ifft_batch,
    lde_batch,
    commit_values,
    hash_rows,
    field_selftest,
    lib_path,
    load_library,
)
from .synth import make_circuit, synth_lib_path  # noqa: F401
from . import parallel  # noqa: F401,E402
from . import translate  # noqa: F401,E402
from . import acir  # noqa: F401,E402
