"""Test infrastructure only: CPU oracle for the CCEdit denoising hot path (see ccedit_oracle.py)."""
