"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (torch-CPU / NumPy) of the reference's GAIL/AIRL round. Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package, and
only as the checker / timed baseline. `imitation_amd/` never imports it.

Pinning status
--------------
* `oracle.imitation_restated` (everything the imitation repo itself owns on the path):
  PINNED -- validated against the reference's own modules executed under `oracle.ref_shim`
  (golden vectors in `tests/golden/*.npz`, generator `tests/golden/make_golden.py`) and
  against the reference's known-answer tests (SURVEY 8c list).
* `oracle.sb3_restated` (PPO / GAE / ActorCriticPolicy -- third-party stable-baselines3
  ~=2.2.1, absent from /root/reference): PARITY UNPINNED numerically; only the fixture
  `model.zip` state-dict layout / Adam eps are pinned.
"""
