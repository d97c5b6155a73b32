"""Host-side mirror of the reference's module API for the hot path, over libzolt_gpu.so.

Names, argument meaning and error behaviour follow the Zig modules so the parity tests
read like the reference's own tests (paths under /root/reference):

  MSM.compute / BatchMSM / ParallelMSM        src/msm/mod.zig:345-748
  HyperKZG.setup / commit / batchCommit / open src/poly/commitment/mod.zig:174-324,558-570
  EqPolynomial.evals, DensePolynomial          src/poly/mod.zig:23-323
  Sumcheck.Prover / Verifier, runSumcheck      src/subprotocols/mod.zig:18-354

All heavy arithmetic runs in the HIP kernels. What stays on the host is exactly what
stays on the host in the reference integration: the toy verifier's 64-bit challenge mixer
and a handful of scalar field operations per round (Python ints below), i.e. the role the
unchanged Zig `field` module plays above the FFI seam.

Field elements are numpy uint64[4] Montgomery limbs; points numpy uint64[8] + inf flag.

The package is split per family (round-3 review): _base (scalars), msm, commitment, wire, poly, sumcheck, transcript,
witness (integer trace columns -> the device-resident witness matrix), provers, blake2b. Every name of every part — the underscore helpers that bench.py and the tests use included — is re-exported
here, so `from zolt_amd import api; api.X` is unchanged.
"""
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403
from .wire import *  # noqa: F401,F403
from .poly import *  # noqa: F401,F403
from .sumcheck import *  # noqa: F401,F403
from .transcript import *  # noqa: F401,F403
from .witness import *  # noqa: F401,F403
from .provers import *  # noqa: F401,F403
from .blake2b import *  # noqa: F401,F403
