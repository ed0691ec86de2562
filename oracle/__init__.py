"""CPU oracle for the PULSE data-parallel RL hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pulse_amd/`` may import, call, link
or execute anything in this package; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg do, and there only as the checker.

Contents
--------
* ``shim/isaacgym/torch_utils.py`` -- file-backed restatement of the absent
  third-party ``isaacgym.torch_utils`` (xyzw quaternion helpers).  Parity of
  THAT boundary is unpinned (no copy of Isaac Gym and no reference test exists,
  SURVEY.md section 8c).
* ``refload.py`` -- loads the reference's own TorchScript functions from
  ``/root/reference`` (this container only) so that golden vectors can be
  generated from the real implementation (``gen_golden.py``).
* ``rotations.py`` / ``env_oracle.py`` / ``motion_oracle.py`` / ``agent_oracle.py``
  -- the plain PyTorch-CPU restatement the HIP path is compared with.  Each
  function cites the reference file:line it follows.  Pinned (a) against
  ``tests/golden/*.npz`` (outputs of the real reference functions run here) and
  (b) bit for bit against the reference's own METHOD BODIES, AST-extracted by
  ``refload.py`` and executed on stub objects (tests/test_oracle_vs_reference_learning.py,
  tests/test_oracle_env_vs_reference_methods.py, tests/test_teacher.py,
  tests/test_checkpoint_format.py, tests/test_replay_buffer.py).
* What stays restated WITHOUT a pin: the pieces of the absent third-party packages
  (isaacgym.torch_utils; rl_games 1.1.4: the Gaussian model wrapper, policy_kl,
  AverageMeter, ExperienceBuffer, swap_and_flatten01, A2CBase.env_step) -- listed in
  DESIGN.md section 4.
"""
