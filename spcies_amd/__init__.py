"""spcies_amd - MI355X-native batched MPC solve engine (HIP platform of the Spcies solver family).

Host side mirrors the reference's generator interface (``spcies_gen_controller`` and the generated
``[u, k, e_flag, sol] = solver(x0, xr, ur)`` call); compute runs in ``libspcies_hip.so`` (hand-written
gfx950 HIP kernels behind the C-ABI of ``include/spcies_hip.h``).
"""
__version__ = "0.1.0"
