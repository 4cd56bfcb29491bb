#!/usr/bin/env python3
"""CLI of the circuit drivers: python tools/run_circuit.py stage1|stage2|stage3 ... (see drivers.main)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcr_amd
sys.exit(pcr_amd.drivers.main())
