/* pgmock stand-in for postgres.h (TEST INFRASTRUCTURE ONLY, see ../pgmock.h) */
#include "pgmock.h"
