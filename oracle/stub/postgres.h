/* TEST INFRASTRUCTURE ONLY (oracle/).  Minimal stand-in for the server header
 * that distfunc.c:15 includes; the hot-path files only need the C99 integer,
 * bool and size types from it.  No Postgres API is provided. */
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
