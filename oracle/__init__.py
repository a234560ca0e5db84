"""Test infrastructure: CPU oracle for the torecsys hot path (see cpu_ref.py). Never imported by torecsys_amd."""
