#!/bin/bash
timeout 300 python scripts/prof_ddpg_host.py 2>&1 | grep -v "^$" | tail -45 | cut -c1-170
