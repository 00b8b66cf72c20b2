"""CPU oracle for the myrtlespeech hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``myrtlespeech_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker, never as the thing measured or shipped.
"""
