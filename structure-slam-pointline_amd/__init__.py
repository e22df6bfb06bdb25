"""MI355X-native point+line feature front-end (Structure-SLAM drop-in).
The directory name carries a hyphen, so load it with tests/pkg.py (importlib)."""
