"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement of the reference's `pointnet2_batch_cuda` kernels, used only as
the checker in tests/, in `__graft_entry__.smoke()` and as the timed CPU
baseline in `bench.py`.  Nothing under `adaptpoint_amd/` or the drop-in
`pointnet2_batch_cuda.py` imports this package.
"""
