"""montecarlo.jl_amd — MI355X-native DQMC sweep engine behind the plugin surface of
ffreyer/MonteCarlo.jl's DQMC flavor (src/flavors/DQMC + src/linalg).

Python here is host plumbing only: lattice/model construction, loop control and the
ctypes binding of libdqmc_hip.so.  All numerics run in hand-written gfx950 kernels;
there is no CPU fallback (importing without the built library raises)."""
from ._lib import DQMCError, lib  # noqa: F401
from .configurations import (CompressedConf, ConfigRecorder, Discarder, compress,  # noqa: F401
                             decompress)
from . import lattices  # noqa: F401
from .lattices import (Chain, EachLocalQuadByDistance, EachSitePairByDistance, SquareLattice,  # noqa: F401
                       build_checkerboard)
from .models import (HubbardModel, HubbardModelAttractive, HubbardModelRepulsive,  # noqa: F401
                     rand_conf)
from .sharding import (Communicator, reduce_accumulators, walker_block, walker_range,  # noqa: F401
                       walker_seeds)
from .dqmc import (DQMC, DQMCParameters, calculate_greens_AVX, device_count,  # noqa: F401
                   checkerboard_exponentials, checkerboard_tables, hopping_exponentials, mfma_f64_peak, rdivp, udt_AVX_pivot, vmul)

lib()  # fail loudly at import time if the HIP library has not been built
