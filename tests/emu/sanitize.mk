# CPU-ONLY test infrastructure: the emulation build of the engine (tests/emu/Makefile: the product's kernel source
# compiled by g++ with -DCO_EMU) under g++'s address and undefined-behaviour sanitizers.
#
#     make -C tests/emu -f sanitize.mk asan          # build + run the emulation-build tests under the sanitizers
#     make -C tests/emu -f sanitize.mk asan TESTS="tests/test_tourney.py"
#
# This file is listed in .gpurunignore: it stays in the build container (GPU sanitizer / XNACK runs are not
# available on the GPU pool; nothing here touches a GPU).
include Makefile
SAN_LIB = /tmp/libcorintho_emu_san.so
SAN_FLAGS = -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer
TESTS ?= tests/test_engine_parity.py tests/test_resident_pool.py tests/test_analyse.py tests/test_game_logs.py tests/test_tourney.py
ASAN_RT := $(shell $(CXX) -print-file-name=libasan.so)
UBSAN_RT := $(shell $(CXX) -print-file-name=libubsan.so)

$(SAN_LIB): $(DEPS)
	$(CXX) $(CXXFLAGS) $(SAN_FLAGS) -shared -o $@ -x c++ $(CSRC)/engine.hip -x c++ nn_emu.cpp -lm

asan: $(SAN_LIB)
	cd ../.. && CO_EMU_LIB=$(SAN_LIB) LD_PRELOAD="$(ASAN_RT) $(UBSAN_RT)" \
	  ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	  python -m pytest $(TESTS) -x -q -m "not gpu" -p no:cacheprovider
.PHONY: asan
