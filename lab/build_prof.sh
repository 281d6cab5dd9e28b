#!/bin/bash
# diagnostic build: the product objects + ft_rnn_persist.hip compiled with -DFT_RNN_PROF (s_memtime phase stamps)
# -> lab/libfwdtaco_prof.so, used through FT_LIB=lab/libfwdtaco_prof.so by lab/rnn_phase_prof.py
set -e
cd "$(dirname "$0")/../forwardtacotron_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DFT_RNN_PROF -I ../../include -I . -c ft_rnn_persist.hip -o /tmp/ft_rnn_persist_prof.o
objs=$(ls build/*.o | grep -v ft_rnn_persist.o)
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -Wl,-z,defs -o ../../lab/libfwdtaco_prof.so $objs /tmp/ft_rnn_persist_prof.o
echo built lab/libfwdtaco_prof.so
