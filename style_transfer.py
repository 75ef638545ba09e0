#!/usr/bin/env python3
"""Entry point with the reference's name: ``python style_transfer.py -ci C -si S [options]``."""
import sys

from style_transfer_amd.cli import main

if __name__ == '__main__':
    sys.exit(main())
