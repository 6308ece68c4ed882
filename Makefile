# MI355X-native conjugate gradient: the library, the C driver and the test oracle.
# (The same commands as conjugate-gradient_amd/build.py, for people who do not want Python.)
#
#   make          libcg_hip.so + src/cg (+ src/cg_mpi when MPI_PREFIX has an MPI)
#   make oracle   CPU restatement used by the tests (oracle/), never linked into the product
#   make clean

HIPCC      ?= hipcc
CC         ?= gcc
MPI_PREFIX ?= /opt/conda
PKG        := conjugate-gradient_amd
LIB        := $(PKG)/libcg_hip.so
# -ffp-contract=off: the reference rounds every product before the following add; the default would fuse them.
HIPFLAGS   := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I/opt/rocm/include -Wall -Wno-unused-function
CFLAGS     := -std=c99 -D_POSIX_C_SOURCE=200809L -O2 -Wall -Iinclude
LINK       := -L$(PKG) -lcg_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -lm
HAVE_MPI   := $(wildcard $(MPI_PREFIX)/lib/libmpi.so)

all: $(LIB) src/cg $(if $(HAVE_MPI),src/cg_mpi)

$(LIB): $(wildcard $(PKG)/csrc/*) include/cg_hip.h
	$(HIPCC) $(HIPFLAGS) $(PKG)/csrc/cg_lib.hip -o $@ -ldl

src/cg: src/cg.c src/utils_hip.c include/utils.h include/cg_hip.h $(LIB)
	$(CC) $(CFLAGS) src/cg.c src/utils_hip.c -o $@ $(LINK)

# one rank per GPU under mpirun; the compiler is called directly (an mpicc wrapper's -L would let the
# linker resolve the HIP runtime's libstdc++ against the MPI distribution's older copy)
src/cg_mpi: src/cg.c src/utils_hip.c include/utils.h include/cg_hip.h $(LIB)
	$(CC) -DCG_USE_MPI -I$(MPI_PREFIX)/include $(CFLAGS) src/cg.c src/utils_hip.c -o $@ $(LINK) \
	  $(MPI_PREFIX)/lib/libmpi.so -Wl,--enable-new-dtags -Wl,-rpath,$(MPI_PREFIX)/lib -Wl,--allow-shlib-undefined

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(LIB) src/cg src/cg_mpi
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
