// TEST INFRASTRUCTURE (oracle/ref_shim): regenie reads optional .bgi index files through sqlite3
// (src/Geno.cpp:180-385).  The library is absent here; these stand-ins make the calls fail cleanly
// ("cannot open"), so a run that asks for a .bgi stops with regenie's own error path.
#ifndef RG_SHIM_SQLITE3_HPP
#define RG_SHIM_SQLITE3_HPP
struct sqlite3; struct sqlite3_stmt;
#define SQLITE_OK 0
#define SQLITE_ROW 100
#define SQLITE_DONE 101
inline int sqlite3_open(const char*, sqlite3** db) { *db = 0; return 1; }
inline const char* sqlite3_errmsg(sqlite3*) { return "sqlite3 is not available in this build (oracle/ref_shim)"; }
inline int sqlite3_prepare_v2(sqlite3*, const char*, int, sqlite3_stmt** s, const char**) { *s = 0; return 1; }
inline int sqlite3_step(sqlite3_stmt*) { return SQLITE_DONE; }
inline const unsigned char* sqlite3_column_text(sqlite3_stmt*, int) { return (const unsigned char*)""; }
inline int sqlite3_finalize(sqlite3_stmt*) { return 0; }
inline int sqlite3_close(sqlite3*) { return 0; }
#endif
