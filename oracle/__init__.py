"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the StemGNN spectral hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.
"""
