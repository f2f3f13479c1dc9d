"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm for the detect/track -> 2D -> 3D hot path, used as the
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
posepipeline_amd/ may import this package: the product path is the HIP library and fails loudly
without it.
"""
