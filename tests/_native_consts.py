# record layout constants, mirrored from include/marlgrid_hip.h for the tests
AG_X, AG_Y, AG_DIR, AG_FLAGS, AG_CARRY, AG_RANK, AG_BONUS = range(7)
AF_ACTIVE, AF_DONE, AF_PLACED, AF_EVICTED = 1, 2, 4, 8
