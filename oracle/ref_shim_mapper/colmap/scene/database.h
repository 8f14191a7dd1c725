#pragma once
// Stand-in (test infrastructure, oracle/Makefile ref_mapper): glomap/controllers/global_mapper.h names colmap::Database in the
// signature of GlobalMapper::Solve; only the retriangulation step (skipped here: it is COLMAP's incremental mapper) reads it.
namespace colmap {
class Database {};
}  // namespace colmap
