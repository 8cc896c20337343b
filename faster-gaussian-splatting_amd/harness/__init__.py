"""Minimal stand-ins for the NeRFICG pieces the FasterGS hot path is called from:
synthetic scenes/views (scenes.py), the Gaussians container + one-view training step (trainer.py),
and the view-parallel multi-GPU step (distributed.py)."""
