#pragma once
// Stand-in: global_mapper.cc includes <colmap/util/file.h> and uses nothing of it.
