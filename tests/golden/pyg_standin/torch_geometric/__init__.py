"""Stand-in for torch-geometric==1.2.1, used ONLY by tests/golden/make_golden.py in the build
container to import the reference's model/network.py (PyG is not installable here).  Semantics are
the oracle's restatement (oracle/dense_ref.py) -- parity is unpinned at this boundary."""
