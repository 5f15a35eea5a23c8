"""gigl_amd — MI355X-native k-hop subgraph sampler + GNN aggregation path for GiGL.

Host-side mirror of the reference's interfaces for this one hot path; all compute
goes through the C ABI in include/gigl_hip.h (gigl_amd/csrc -> libgigl_hip.so).  There is no CPU
fallback: importing the package works anywhere, calling a compute op without the HIP library and a
GPU raises.
"""
__version__ = "0.1.0"
