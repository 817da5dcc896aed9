#!/usr/bin/env python3
"""Drop-in for the reference's tools/refinement.py (same argv, files and exit codes);
the work is done by beyond_fixed_forms_amd on an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd.cli import refinement_main  # noqa: E402

if __name__ == "__main__":
    sys.exit(refinement_main())
