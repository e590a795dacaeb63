"""TEST INFRASTRUCTURE ONLY.

`oracle.cpu`  -- ctypes binding of oracle/liboracle.so, our scalar C restatement of
                 the reference's hot-path algorithms (oracle/gdsp_oracle.c).
`oracle.ref`  -- ctypes binding of oracle/_ref/libgenodsp_ref.so, the unmodified
                 reference compiled from /root/reference by `make -C oracle ref`
                 (present only where that build was possible; `ref.available()`).

Nothing under genodsp_amd/ imports this package: the product path is the HIP
library and fails loudly without it.  Importers are tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke().
"""
