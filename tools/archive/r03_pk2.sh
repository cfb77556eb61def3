#!/bin/bash
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/pk_opsel_repro tools/pk_opsel_repro.hip && /tmp/pk_opsel_repro
