"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the hcatgnet GCN hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the timed CPU baseline -- never as a compute path of ``hcatgnet_amd``.
"""
