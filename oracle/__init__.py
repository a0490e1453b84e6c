"""CPU oracle for the osu-diffusion DiT denoising path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker — never as the thing measured or shipped.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(`/root/reference`, Python) in the build container, checks this restatement against
it and freezes input/output vectors into ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` re-checks the oracle against those vectors on any
machine (no reference needed).
"""
