from oracle.dense_ref import DenseSAGEConv, DenseGINConv  # noqa: F401
