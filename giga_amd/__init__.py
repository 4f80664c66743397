"""giga_amd: MI355X-native (gfx950) implementation of GIGA's dense inference path.

    from giga_amd.networks import get_network, load_network     # vgn.networks drop-in (GIGA entries)
    from giga_amd.detection import query_lattice, predict       # vgn.detection_implicit counterpart

All arithmetic runs in hand-written HIP kernels behind the C ABI of include/giga_hip.h."""
__version__ = "0.1.0"
