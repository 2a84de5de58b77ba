"""CPU oracle for the CGC-Net hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``cgc-net_amd/`` imports it
and the product path raises when the HIP library is missing.

Contents
--------
``dense_ref``  dense restatement (torch, CPU, fp32) of the reference algorithm:
               densify -> DenseSAGEConv x15 -> BatchNorm over B*Nmax rows ->
               DiffPool x2 -> max readout -> MLP -> CE.  Follows
               ``model/network.py`` and ``model/utils.py`` of the reference
               (file:line cited per function).
``flat_ref``   the same maths re-expressed on the flat / CSR layout the HIP
               kernels use, one function per C-ABI op, so that each kernel has
               an op-level checker (and autograd gives its backward).

Pinning
-------
* Everything the reference repository computes itself is pinned: the golden
  fixtures under ``tests/golden`` were produced by importing the reference's
  ``model/network.py`` in the build container (script: ``tests/golden/make_golden.py``).
* PARITY UNPINNED at the torch_geometric boundary: ``DenseSAGEConv``,
  ``DenseGINConv``, ``to_dense_batch`` and ``scatter_`` live in
  torch-geometric==1.2.1 (requirements.txt:40 of the reference), which is not
  installable here and for which the reference holds no tests or golden
  vectors.  Their published semantics are restated in ``dense_ref`` and the
  same restatement backs the stand-in used to import the reference.
"""
