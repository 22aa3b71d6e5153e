// placement.cpp -- where a handle's host threads run: the NUMA node of its GPU and the CPUs of that node inside the process's
// affinity mask (SURVEY 8e names the host as the limiter of eight GPUs on one box; DESIGN 5, `host`).  Moved out of model.cpp in round 6.
#include <sched.h>

#include <cctype>
#include <fstream>

#include "model_types.hpp"
#include "model_parts.hpp"

namespace ufd {
// ---------------------------------------------------------------- host placement
// "0-3,8,10-11" -> cpu ids
std::vector<int> parse_cpu_list(const std::string& txt) {
  std::vector<int> out;
  size_t i = 0;
  while (i < txt.size()) {
    while (i < txt.size() && !std::isdigit((unsigned char)txt[i])) i++;
    if (i >= txt.size()) break;
    int a = 0;
    while (i < txt.size() && std::isdigit((unsigned char)txt[i])) a = a * 10 + (txt[i++] - '0');
    int b = a;
    if (i < txt.size() && txt[i] == '-') {
      i++;
      b = 0;
      while (i < txt.size() && std::isdigit((unsigned char)txt[i])) b = b * 10 + (txt[i++] - '0');
    }
    for (int c = a; c <= b && out.size() < 4096; c++) out.push_back(c);
  }
  return out;
}

std::string read_first_line(const std::string& path) {
  std::ifstream f(path);
  std::string line;
  if (f) std::getline(f, line);
  return line;
}

// NUMA node of the device (/sys/bus/pci/devices/<bdf>/numa_node) and the CPUs of that node inside this process's
// affinity mask.  Nothing is pinned when the node is unknown (-1: one socket, or a VM that hides the topology), when
// the mask and the node do not intersect, or with UFD_FLAG_NO_NUMA_PIN.
void resolve_placement(ufd_model* m) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), m->cfg.device_id) != hipSuccess) return;
  for (char* p = bdf; *p; p++) *p = (char)std::tolower((unsigned char)*p);
  m->pci_bdf = bdf;
  const std::string node_txt = read_first_line(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
  if (node_txt.empty()) return;
  m->numa_node = std::atoi(node_txt.c_str());
  if (m->numa_node < 0 || (m->cfg.flags & UFD_FLAG_NO_NUMA_PIN)) return;
  const std::vector<int> node_cpus =
      parse_cpu_list(read_first_line("/sys/devices/system/node/node" + std::to_string(m->numa_node) + "/cpulist"));
  cpu_set_t cur;
  CPU_ZERO(&cur);
  if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
  for (int c : node_cpus)
    if (c < CPU_SETSIZE && CPU_ISSET(c, &cur)) m->pin_cpus.push_back(c);
  // compact "a-b,c" form for reports
  std::string txt;
  for (size_t i = 0; i < m->pin_cpus.size();) {
    size_t j = i;
    while (j + 1 < m->pin_cpus.size() && m->pin_cpus[j + 1] == m->pin_cpus[j] + 1) j++;
    txt += (txt.empty() ? "" : ",") + std::to_string(m->pin_cpus[i]) + (j > i ? "-" + std::to_string(m->pin_cpus[j]) : "");
    i = j + 1;
  }
  m->cpu_list = txt;
}

// Calling thread -> the handle's CPUs (no-op when nothing was resolved).
void pin_this_thread(const ufd_model* m) {
  if (m->pin_cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : m->pin_cpus) CPU_SET(c, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}
}  // namespace ufd
