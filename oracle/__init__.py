"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch-CPU fp32 / numpy fp64, no HIP, no nn.Module tree) of the
volumetric-triangulation forward path of karfly/learnable-triangulation-pytorch, used as the
parity checker for the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here; the product package
(``learnable-triangulation-pytorch_amd/``) never does.

Pinning: the reference publishes no golden vectors or tests for this path, so
this oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, executed in the build container by
``oracle/make_golden.py`` (imports /root/reference with two stub modules) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks oracle == golden on every run.
"""
