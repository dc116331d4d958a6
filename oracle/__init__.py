"""CPU restatement of the reference's pruning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under spice_amd/ imports this package; it is used by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker the
HIP path is compared with.  Every function cites the reference file:line it restates.

Pinning: the reference ships no tests or golden vectors.  The functions restating in-tree
reference code (losses, Taylor scores, quota allocation, confusion/IoU, importance plug-ins)
are pinned by fixtures generated from the reference itself (tests/golden/make_golden.py).
The functions restating torch-pruning 1.6.0 / torchmetrics 1.8.1 semantics (magnitude
importance normalisation, channel selection, physical slicing, macro Jaccard) follow the
published algorithms of those libraries, whose sources are absent here: parity unpinned.
"""
