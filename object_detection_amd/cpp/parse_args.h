// Command line of the demo programs: the contract of the reference's
// common/parse_args.h:7-110 (-h | -i image | -v video | -c | -m a b, plus the
// mandatory -f offload with offload in [0, 1] or -1).
#pragma once
#include <cstdlib>
#include <ostream>
#include <string>

namespace iargs
{
    struct app_args
    {
        float offload = 0.0f;
        std::string app_mode{}, img_path_a{}, img_path_b{};

        static bool usage(std::ostream& stream)
        {
            stream << "Usage exmaple: app.exe -h" << std::endl;
            stream << "Usage exmaple: app.exe -i image_path<string> -f offload<float>" << std::endl;
            stream << "Usage exmaple: app.exe -v video_path<string> -f offload<float>" << std::endl;
            stream << "Usage exmaple: app.exe -c -f offload<float> (camera_IO mode, try use sensor)" << std::endl;
            stream << "Usage exmaple: app.exe -m img_path_a<string> img_path_b<string> -f offload<float>" << std::endl;
            return false;
        }

        void dump(std::ostream& stream) const
        {
            stream << "App_mode:\t" << app_mode << std::endl;
            stream << "Path_img_a:\t" << img_path_a << std::endl;
            stream << "Path_img_b:\t" << img_path_b << std::endl;
            stream << "Offload:\t" << offload << std::endl;
        }

        bool parse(int argc, char** argv)
        {
            bool have_offload = false;
            auto arg = [&](int i) { return i < argc ? std::string(argv[i]) : std::string(); };
            for (int i = 0; i < argc; i++) {
                const std::string str = argv[i];
                if (str == "-i" || str == "-v") {
                    app_mode = str;
                    img_path_a = arg(i + 1);
                } else if (str == "-h" || str == "-c") {
                    app_mode = str;
                } else if (str == "-m") {
                    app_mode = str;
                    img_path_a = arg(i + 1);
                    img_path_b = arg(i + 2);
                } else if (str == "-f") {
                    have_offload = true;
                    offload = (float)std::atof(arg(i + 1).c_str());
                }
            }
            if (app_mode == "-h") return true;
            if (!have_offload || app_mode.empty()) return false;
            if (offload != -1.0f && (offload < 0.0f || offload > 1.0f)) return false;
            if ((app_mode == "-i" || app_mode == "-v") && img_path_a.empty()) return false;
            if (app_mode == "-m" && (img_path_a.empty() || img_path_b.empty())) return false;
            return true;
        }
    };
}
