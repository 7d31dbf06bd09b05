/* pgmock: every Postgres declaration the reference glue needs lives in pgmock.h (TEST INFRASTRUCTURE ONLY) */
#include "pgmock.h"
